// Host-side state of libbsr_hip.so shared by its translation units: the context, a batch slot, error plumbing, the
// staging entry points.  bsr_api.hip: C ABI and the batch pipeline; bsr_stage.hip: tapes -> streams, tape groups, launch
// records; bsr_place.hip: CPU placement of the library's threads; bsr_comm.hip: the RCCL gather.
#pragma once
#include <rccl/rccl.h>

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "bsr_internal.h"
#include "bsr_span.h"

#define BSR_HID __attribute__((visibility("hidden")))

extern thread_local std::string g_create_error;

struct TapeLoc {
  int code_off, n_nodes, feat_off, ln_off, max_sp;
  int n_stream;  // opcode-stream entries after fusing `terminal, +|*` pairs
  int cost;      // rough relative cost of one sweep of the tape (orders the work queue: heaviest first)
  int nt, nl;      // terminals in the column stream, (a, b) pairs in the ln stream
  int grp;         // tile pass: the tape group that runs the tape (its LDS slot map numbers the tape's columns)
  int acc_only;    // chain tape: one leading terminal, every other stream entry maps the accumulator to itself
};

// Everything one batch in flight owns.  Two slots let the host stage batch i+1 while the GPU scores batch i.
// The input block is one pinned host buffer mirrored by one device buffer and uploaded with a single copy:
//   [ feature list (d int32) | descriptors (max_batch + 1 PropDesc) | opcode words | column words | ln pairs ]
struct BatchSlot {
  uint8_t* h_in = nullptr;
  uint8_t* d_in = nullptr;
  size_t in_cap = 0;
  size_t off_cols = 0, off_sched = 0, off_mh = 0, off_desc = 0, off_streams = 0;
  int cols_stride = 0;   // entries between the tape groups' column-pointer tables
  size_t off_recs = 0;   // tile schedule (tape records, then the cost-order index): behind the batch's streams
  size_t recs_bytes = 0; // ... of which this batch uses so many bytes
  size_t srec_off = 0;   // streaming kernel: its StreamRecs, so many bytes behind off_recs
  bool tile_deep2 = false;           // ... this batch takes the streaming kernel with a second value below the accumulator
  std::vector<uint64_t> ext_words;   // ... and the extension words of the batch's long programs (scratch of build_tile_launch)
  int32_t stat_tapes = 0, stat_fast = 0, stat_chain = 0, stat_entries = 0;   // bsr_batch_stats: the last batch staged here
  size_t tprog_off = 0;  // whole-slice kernel with the assembly tape loop: its TileProgs, so many bytes behind off_recs (0: none)
  // device-side MH step (bsr_score_submit_mh): per-proposal terms and flags, span offsets; results
  MhRes* d_mh = nullptr;
  bsr_event* h_ev = nullptr;   // pinned, written by k_events
  int n_spans = 0;             // spans of the batch in flight (0: plain scoring)
  size_t mh_cap = 0;           // proposals the MH block has room for
  size_t code_words = 0, feat_words = 0, ln_words = 0;
  // tile pass (bsr_tile.hip): a second column stream with LDS slots instead of X columns sits behind the ln pairs
  bool tile = false;
  bool tile_possible = false;   // stage_tapes: nothing rules the tile pass out for the staged batch (stage_tile decides)
  int tile_chains = 0;                // distinct chains of the batch (their basis columns are staged in LDS)
  int tile_group_chains = 0;          // ... per tape group (= tile_chains unless the groups are chain groups)
  std::vector<int32_t> chain_slot;    // chain -> index among the batch's chains, -1 if absent
  std::vector<int32_t> batch_chains;  // the batch's chains in first-seen order
  std::vector<double> wave_load;      // scratch of the tile schedule
  std::vector<int> wave_cnt;
  bsr_score* h_out = nullptr;
  hipStream_t stream = nullptr;   // each slot has its own stream: the small solve/residual/finalise kernels of one
                                  // batch overlap the row pass of the other
  PropCoef* d_coef = nullptr;
  int32_t* d_flagged = nullptr;  // two lists ([0] count, [1..] proposals k_solve hands to the residual pass), alternating by
                                 // batch: each batch's k_solve empties the list the batch before it used
  int flag_par = 0;
  int32_t* flag_cur() const { return d_flagged + (size_t)flag_par * flag_stride; }
  int32_t* flag_other() const { return d_flagged + (size_t)(flag_par ^ 1) * flag_stride; }
  size_t flag_stride = 0;
  double* part1 = nullptr;
  double* part2 = nullptr;
  uint32_t* h_done_word = nullptr;   // pinned: the batch's completion, written by the queue behind its last kernel (hipStreamWriteValue32)
  uint32_t done_gen = 0, done_wanted = 0;
  AqlSlot aql;                       // direct dispatch (bsr_aql.h): completion signal, kernel-argument block, queue
  std::unique_ptr<AqlBatch> aqb;     // ... the packets of the batch being issued
  int aql_items = 0;                 // ... how many packets it had
  bool aql_pending = false;          // ... the batch in flight went that way: the waiter polls the signal   // ... the value of the batch in flight (0: the event `done` is what the waiter blocks on)
  bool part2_uncached = false;   // part2 and the arrival counter behind it are uncached memory: the finalise step is fused
  size_t part_cap = 0;   // in (proposal,row block) records
  void* spill = nullptr;
  size_t spill_cap = 0;  // bytes
  int32_t* queue = nullptr;  // ring of work-queue counter sets for the projection pass (zeroed once; every launch
                             // takes the next set and clears the one half a ring ahead)
  uint32_t queue_seq = 0;
  hipEvent_t done = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int P = 0;
  int nF = 0;
  int derived_used = 0;  // derived columns the batch being staged refers to
  bool use_lds = false;
  int rb_rows = 512;
  bool pending = false;
  uint32_t gen = 0;         // bumped by every submission on this slot
  uint32_t waited_gen = 0;  // generation whose results the last wait on this slot handed out
  // submission thread (see Launcher): the generation whose solve / residual / finalise launches and `done` event have
  // been issued, and the status of issuing them
  std::atomic<uint32_t> tail_gen{0};
  uint32_t tail_wanted = 0;
  int tail_rc = 0;
  int timed = 0;        // profiling level the pending batch was enqueued with
  bool scored = false;  // holds a scored batch (bsr_commit may re-run its tapes)
  std::vector<int32_t> slot_of;  // feature -> LDS slot of the batch being staged
  std::vector<std::pair<int, int>> derived_cand;   // scratch of the derived-column choice
  std::vector<int> derived_benefit;
  // tile pass: per tape group, the columns its tapes read (ascending) -> LDS slots 0.., then y, then the chains' bases
  std::vector<int16_t> grp_slot;     // [group][column] LDS slot or -1
  std::vector<int32_t> grp_cols;     // scratch: columns of one group
  int grp_nF[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // X columns per group (= the group's y slot)
  int tile_ncols = 0;                // most LDS columns of any group: sizes the LDS buffers
  int tile_chunk = 0;                // blocks staged at a time: the whole slice, or a chunk of the ring
  int tile_ring = 1;                 // LDS buffers the chunks travel through (LDS-DMA)
  bool tile_stream = false;          // the batch takes the streaming kernel (bsr_stream.hip: k_stream), not k_tile
  std::vector<std::shared_ptr<const bsr_span::SpanBasis>> span_snap;   // [chain] the bases this batch was staged against
  uint64_t bar_readback = 0;         // (sink of the read that closes a BAR upload)
  std::vector<bsr_node> rows_copy;   // the scored batch's tapes (a commit makes one of them a current tree)
  std::vector<int32_t> off_copy;
  std::vector<int32_t> sub_chain, sub_k;   // the submitted batch's chain / tree index / sigma per proposal, and the chain's
  std::vector<double> sub_sigma, sub_s;    // prescale for that tree as it stood at the submit (the staging may run later)
  std::vector<int> order_tmp;    // the staged batch's tapes by cost, heaviest first (cost_order)
  std::vector<TapeLoc> loc_tmp;   // scratch of a submission
  std::vector<uint32_t> order_keys;
  int order_n = -1;              // tapes order_tmp is valid for (-1: not)
  std::vector<bsr_node> rows_perm;   // tapes rewritten in fusing order (reorder_tape), at their batch offsets
  std::vector<const bsr_node*> tape_src;   // per tape: where the streams are written from (the caller's rows, or rows_perm)
  std::vector<int32_t> perm_kid, perm_stack;   // scratch of reorder_tape
  bool stream_dirty = false;     // work on the slot's stream that no wait has covered yet (bsr_commit's re-run, a rescore's
                                 // descriptor restore): the next batch's input block then goes by a copy command

  int32_t* h_feat() const { return reinterpret_cast<int32_t*>(h_in); }
  const void** h_cols() const { return reinterpret_cast<const void**>(h_in + off_cols); }
  TapeRec* h_sched() const { return reinterpret_cast<TapeRec*>(h_in + off_recs); }
  double* h_terms() const { return reinterpret_cast<double*>(h_in + off_mh); }
  int32_t* h_mhflags() const { return reinterpret_cast<int32_t*>(h_in + off_mh + mh_cap * 8 * sizeof(double)); }
  int32_t* h_spans() const { return h_mhflags() + mh_cap; }
  const double* d_terms() const { return reinterpret_cast<const double*>(d_in + off_mh); }
  const int32_t* d_mhflags() const { return reinterpret_cast<const int32_t*>(d_in + off_mh + mh_cap * 8 * sizeof(double)); }
  const int32_t* d_spans() const { return d_mhflags() + mh_cap; }
  const void* const* d_cols() const { return reinterpret_cast<const void* const*>(d_in + off_cols); }
  const TapeRec* d_sched() const { return reinterpret_cast<const TapeRec*>(d_in + off_recs); }
  PropDesc* h_desc() const { return reinterpret_cast<PropDesc*>(h_in + off_desc); }
  uint64_t* h_streams() const { return reinterpret_cast<uint64_t*>(h_in + off_streams); }
  const int32_t* d_feat() const { return reinterpret_cast<const int32_t*>(d_in); }
  PropDesc* d_desc() const { return reinterpret_cast<PropDesc*>(d_in + off_desc); }
  const uint64_t* d_streams() const { return reinterpret_cast<const uint64_t*>(d_in + off_streams); }
};

struct bsr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t N = 0, ld = 0;
  int d = 0, K = 0, n_chains = 0, max_batch = 0, dtype = 0;
  int n_cols = 0;          // columns of Xt: the d features, then (derived columns on) d per unary opcode of kDerivedOps
  bool derived_ready = false;
  int derived_max = 8;         // BSR_DERIVED_MAX: cap on the derived columns one batch may use
  bool tile_ever = false;   // some batch of this context can take the tile pass
  int n_cu = 256;
  int tile_cus = 256;  // CUs the tile row pass runs on; the other n_cu - tile_cus serve the small kernels behind it
  int aux_cus = 0;
  int bar_write = 0;       // the host writes a batch's input block straight into device memory (large-BAR devices)
  int wgs_per_cu = 5;  // resident 4-wave workgroups per CU the work-queue row pass is sized for (f64 kernels: 92 VGPRs -> 5)
  size_t esz = 8;
  bool has_y = false;
  void* Xt = nullptr;
  void* y = nullptr;
  void* cur = nullptr;   // [chain][k][ld]
  void* Q = nullptr;     // [chain][K][ld]: one orthonormal basis of the chain's K current columns
  void* zbuf = nullptr;  // [max_batch][ld], allocated on the first bsr_eval_tapes that wants columns
  ChainB* d_ck = nullptr;        // [chain]
  std::vector<ChainB> h_ck;
  ChainFitOut* d_fit = nullptr;  // [chain] no-intercept fit (also carries per-column max/flags) + 1 scratch slot
  std::vector<ChainFitOut> h_fit;
  ChainFitOut* d_fit_icpt = nullptr;   // [chain] intercept fit of the last refresh
  std::vector<ChainFitOut> h_fit_icpt;
  RefreshIn* d_rin = nullptr;          // [chain]
  std::vector<RefreshIn> h_rin;        // host-tracked max|.| and inf/NaN flags of every current column
  RefreshPlan* d_plan = nullptr;       // one scratch plan (refreshes are serialised on the main stream)
  RefreshPlan* h_plan = nullptr;       // pinned
  double* d_rpart = nullptr;
  int fast_refresh = 1;
  std::vector<double> x_lo, x_hi;  // per-feature range of X (host side; the native sampler's rank-gate predictor)
  std::vector<char> ready;       // chain factors valid
  std::vector<char> col_set;     // [chain*K+k] column initialised
  // structure of the chains' current trees (bsr_span.h): per tree its linear form, per chain the echelon basis of its K
  // forms -- rebuilt on set_current / commit, handed to the batches in flight as immutable snapshots
  std::vector<bsr_span::LinForm> cur_form;     // [chain*K+k]
  std::vector<uint64_t> cur_fmask;   // features each current tree reads (bit f mod 64), all ones: unknown
  std::vector<char> cur_form_ok;               // ... valid
  std::vector<std::shared_ptr<const bsr_span::SpanBasis>> span;   // [chain]
  int stream_deep = 2;                         // BSR_STREAM_DEEP: the streaming pass's chunk block also takes programs of several words (1) and, K <= 3, a second value below the accumulator (2); 0: round 5's limits
  int solve_exact = 0;                         // BSR_SOLVE_EXACT=1: every proposal's singular values by Jacobi (round 5's k_solve; the standalone ylogLike)
  int selfdup = 1;                             // BSR_SELFDUP: recognise proposals that repeat the tree they replace
  int reorder = 1;      // BSR_REORDER: commutative operands in fusing order (reorder_tape)
  int chain_eval = 1;   // BSR_CHAIN_EVAL: chain tapes take the register-resident pass of the tile kernel
  BatchSlot slot[BSR_SLOTS];
  int next_slot = 0;
  int last_waited = -1;
  std::mutex mu;  // commit / refresh / fit share the main stream and one set of staging buffers (bsr_internal_lock)
  struct Launcher* launcher = nullptr;  // second submission thread (BSR_SUBMIT_THREAD, default on)
  std::mutex err_mu;  // the error text may be written by worker threads
  double* d_stage = nullptr;  // fp64 staging for column download in f32 mode
  // tuning
  int rb_rows = 512;
  int target_wgs = 2048;
  int rows_per_lane = 2;
  int no_lds = 0;
  // tile pass geometry, fixed for the life of the context (a proposal's partial sums must not depend on the batch)
  int tile_on = 1;
  int tile_T = 1, tile_slices = 256, tile_bps = 1, tile_blocks = 1, tile_left = 0;
  bool tile_by_chain = false;   // the tape groups are chain groups: a group stages its own chains' basis columns only (bsr_api.hip geometry)
  int tile_qmax = 4;        // sets of sums per wave (tile_qmax(K))
  int tile_asm = 0;         // whole-slice fp64 contexts of K <= 4: the tape loop in assembly (bsr_tile_asm.hip; BSR_TILE_ASM=0: k_tile1)
  int tile_split = 1;       // ... its staging in two halves (BSR_TILE_SPLIT=0: everything at the first barrier)
  AqlDevice* aql = nullptr;   // direct AQL dispatch of scoring batches (BSR_AQL=0: HIP launches on the slots' streams)
  // a directly dispatched batch never completed (queue error, or silence for a minute): its packets may still be queued
  // or running and may still write the slot's buffers.  The context refuses further batches and bsr_ctx_destroy frees
  // nothing the GPU can reach (leaked on purpose: a stuck queue must not write into recycled memory)
  std::atomic<bool> poisoned{false};
  std::atomic<bool> aql_off{false};   // ... gave up after a failure (a kernel without descriptor, a queue error); set by any thread
  std::atomic<long long> n_direct{0}, n_streamed{0};   // scoring batches issued either way
  int done_word = 1;        // a batch's completion by a stream write-value into pinned memory, polled (BSR_DONE_WORD=0: event)
  bool tile_whole = false;  // every slice of this context fits LDS whole (staged once); else chunked through two buffers
  bool tile_stream = false; // chunked fp64 context: the streaming kernel (bsr_stream.hip) with its own geometry -- every
  int tile_long = 0;        // block in a slice, the first tile_long slices one block longer, no leftover units
  size_t tile_sched_cap = 0;
  unsigned long long* d_stamps = nullptr;   // BSR_TILE_STAMPS=R: per-wave clock samples of the last R tile launches (a ring)
  int stamp_ring = 0;                       // R
  size_t stamp_block_words = 0;             // one launch's block: [workgroups][16 waves][8] words
  std::atomic<uint32_t> stamp_seq{0};       // tile launches so far (launch n writes block n mod R)
  // profiling: 0 off, 1 events around the row pass only, 2 events around every kernel
  int prof = 0;
  double last_us[5] = {0, 0, 0, 0, 0};
  ncclComm_t comm = nullptr;
  void* comm_buf = nullptr;
  size_t comm_cap = 0;
  std::string err;
};

struct bsr_ctx;
BSR_HID void set_err(bsr_ctx* c, const char* msg);
#define HIPCHK(ctx, call)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      char b_[512];                                                                             \
      snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      set_err((ctx), b_);                                                                       \
      return BSR_E_HIP;                                                                         \
    }                                                                                           \
  } while (0)

// host-side cost of a submission, printed by bsr_ctx_destroy when BSR_HOST_PROF is set
extern std::atomic<long long> g_ns_stage, g_ns_desc, g_ns_enq, g_n_sub;  // worker threads submit too
extern std::atomic<long long> g_ns_issue, g_n_issue, g_ns_wait, g_n_wait;   // HIP calls of a batch; waits
extern const bool g_host_prof;
// BSR_STREAM_STATS=1: what the staged batches hold, by stream opcode and evaluator (printed by bsr_ctx_destroy; the
// instruction audit of the row pass prices a launch with it: tools/isa_audit.py)
extern const bool g_stream_stats;
extern std::atomic<long long> g_ss_entries[2][16], g_ss_tapes[2], g_ss_batches, g_ss_derived, g_ss_cols;
static inline long long host_now() {
  return g_host_prof ? std::chrono::duration_cast<std::chrono::nanoseconds>(
                           std::chrono::steady_clock::now().time_since_epoch()).count()
                     : 0;
}

// Derived columns.  Most transcendental nodes of a proposal sit directly on a terminal (`sin(x3)`, `exp(x0)`), and the
// row passes are bound by the vector instructions those nodes cost -- for every tape and row again.  The context
// therefore keeps op(x_f) for every feature f and every unary opcode without parameters as extra columns behind X
// (computed once, by the same device routines the interpreter runs: the values are bit-identical to an inline
// evaluation), and the stream encoder turns `terminal f, op` into one terminal of column d*(1+m)+f.  What the pass
// pays instead is the read of one more column per distinct (op, feature) of the batch -- bandwidth, which it has.
static const int kDerivedOps[] = {BSR_OP_INV, BSR_OP_NEG, BSR_OP_SIN, BSR_OP_COS, BSR_OP_EXP, BSR_OP_SQUARE, BSR_OP_CUBIC,
                                  BSR_OP_LOG};
static const int kNumDerivedOps = (int)(sizeof(kDerivedOps) / sizeof(kDerivedOps[0]));
static inline int derived_index(int opcode) {
  for (int m = 0; m < kNumDerivedOps; ++m)
    if (kDerivedOps[m] == opcode) return m;
  return -1;
}


BSR_HID int fail(bsr_ctx* ctx, int code, const char* msg);
BSR_HID int env_int(const char* name, int dflt);
BSR_HID void poison(void* p, size_t bytes);
static inline void* col_ptr(const bsr_ctx* c, void* base, int64_t col) {
  return (char*)base + (size_t)col * c->ld * c->esz;
}

// bsr_stage.hip
BSR_HID bool is_binary_op(int op);
BSR_HID int check_tape(bsr_ctx* c, const bsr_node* t, int len, int* max_sp);
BSR_HID int ensure_input(bsr_ctx* c, BatchSlot& s, size_t stream_words);
BSR_HID int stage_tapes(bsr_ctx* c, BatchSlot& s, const bsr_node* rows, const int32_t* tape_off, int n,
                        std::vector<TapeLoc>* loc_out, int tile_chains = 0);
BSR_HID void stage_tile(bsr_ctx* c, BatchSlot& s, int n);
BSR_HID int build_tile_launch(bsr_ctx* c, BatchSlot& s, int P, TileGeom* tgp);
template <typename CostOf>
inline void cost_order(std::vector<int>& order, std::vector<uint32_t>& keys, int n, const CostOf& cost_of) {
  keys.resize((size_t)n);
  for (int i = 0; i < n; ++i) keys[i] = ((uint32_t)(65535 - std::min(65535, std::max(0, cost_of(i)))) << 16) | (uint32_t)(i & 0xFFFF);
  std::sort(keys.begin(), keys.end());
  order.resize((size_t)n);
  for (int i = 0; i < n; ++i) order[i] = (int)(keys[i] & 0xFFFFu);
}


// bsr_place.hip
extern std::atomic<bool> g_pinned;       // BSR_PIN=1: this process confined itself to the library's CPUs
extern cpu_set_t g_lib_cpus;
extern std::atomic<bool> g_lib_cpus_ok;
BSR_HID void choose_lib_cpus(int device);
extern int g_lib_numa;
