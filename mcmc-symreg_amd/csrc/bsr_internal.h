// Internal declarations shared by the kernel and API translation units of libbsr_hip.so.
// gfx950 (MI355X) only: wave64, no portability layer.
#pragma once

#ifndef BSR_HOST_ONLY   // (the sanitizer build of the host-side sampler compiles without HIP: tests/native/build_san.sh)
#include <hip/hip_runtime.h>
#endif
#include <stddef.h>
#include <stdint.h>

#include "../../include/bsr_hip.h"
#ifndef BSR_HOST_ONLY
#include "bsr_aql.h"
#endif

#define BSR_SLOTS (2 * BSR_MAX_INFLIGHT)   // batch slots of a context: the public tickets use the first BSR_MAX_INFLIGHT, the native
                                          // sampler's worker threads a second one each for the batch they generate ahead
#define BSR_NQ_MAX BSR_MAX_K        // basis columns per chain (one orthonormal column per current tree output)
#define BSR_WAVE 64
#define BSR_WG_WAVES 4              // waves per workgroup in the row-pass kernels
// opcode-stream-only codes (never in a bsr_node): acc = acc + X[:,f] / acc * X[:,f] with f from the column stream
#define BSR_QUEUE_SETS 16                             // ring of counter sets per batch slot
#define BSR_QUEUE_SUB 16                              // ticket counters per XCD (power of two, <= 32)
#define BSR_QUEUE_SET_INTS (8 * BSR_QUEUE_SUB * 32)   // one counter per 128-byte line
#define BSR_SOP_ADD_T 11
#define BSR_SOP_MUL_T 12
#ifndef BSR_ROWS_MIN_WAVES
#define BSR_ROWS_MIN_WAVES 4         // occupancy floor requested from the compiler for the row pass
#endif
#ifndef BSR_REG_STACK
#define BSR_REG_STACK 3             // interpreter stack slots held in VGPRs (plus the accumulator)
#endif
#define BSR_ROW_ALIGN 4096          // device columns are padded to a multiple of this many rows
#define BSR_P1_WORDS 12             // doubles per (proposal,row block) partial of pass 1
#define BSR_P2_WORDS 2              // doubles per (proposal,row block) partial of pass 2

enum { BSR_MODE_SCORE = 0, BSR_MODE_EVAL = 1 };

// Cached factors of one chain: ONE orthonormal basis of its K current columns, each prescaled by its own power of
// two d_j:  O_j d_j = sum_i Q_i R_ij  (R upper triangular).  A proposal that replaces tree k uses R with column k
// removed; see DESIGN.md "rank gate and OLS".  Non-finite columns enter as zero columns (d_j = 0).
struct ChainB {
  double R[BSR_NQ_MAX * BSR_NQ_MAX];     // upper triangular, row-major [row][col]
  double qy[BSR_NQ_MAX];                 // Q^T y
  double yperp2;                         // |y - Q Q^T y|^2, measured directly
  double d[BSR_NQ_MAX];                  // per-column power-of-two prescale (0: column holds inf/NaN)
  double s_k[BSR_MAX_K];                 // prescale of a candidate for tree k: pow2 of max |sibling columns|
  double m_other[BSR_MAX_K];             // max |columns j != k| (unscaled)
  uint32_t flags_k[BSR_MAX_K];           // BSR_F_INF / BSR_F_NAN of the columns j != k
};

// Device-side descriptor of one tape to run in the row passes.
struct PropDesc {
  int32_t code_off;     // first 64-bit word of the tape's opcode stream
  int32_t n_nodes;
  int32_t feat_off;     // first 64-bit word of the tape's terminal-column stream
  int32_t ln_off;       // first (a,b) pair of the tape's ln-parameter stream
  int32_t mode;         // BSR_MODE_*
  int32_t nq;           // basis columns (K), 0 in eval mode
  int32_t k;            // tree index being replaced
  int32_t K;
  int32_t ck;           // index into the ChainB array (chain)
  int32_t spill_need;   // stack slots beyond the register stack
  int32_t max_sp;       // values the fused encoding holds at once (the accumulator included): 1 for a chain
  int32_t order;        // entry i: index of the i-th most expensive tape of the batch (work-queue order)
  int32_t cost;         // host's estimate of the tape's cost per sweep (sort key for `order`)
  int32_t qslot;        // tile pass: LDS slot of the chain's first basis column (K consecutive slots)
  int32_t grp;          // tile pass: the tape group that runs the tape
  int32_t n_term;       // terminals in the tape's column stream
  int32_t n_ln;         // ln nodes ((a, b) pairs in its ln stream)
  int32_t chain;        // the tape is a chain (bsr_device.h: chain_eval): first entry a terminal, every other one acc -> acc
  int32_t self_dup;     // the host found the candidate to be the chain's current tree k again (same canonical form, up to
                        // root negation): its column lies in the span exactly, w = 0 needs no residual pass
  const void* qbase;    // first basis column (nq columns, stride ld)
  void* zout;           // where the candidate column goes (ld values) or nullptr
  double s;             // prescale applied to the candidate column in all accumulations
  double sigma;         // new_sigma
};

// What the solve step hands to the residual pass.
struct PropCoef {
  double c[BSR_NQ_MAX];   // Q^T (s z)               : projection of the candidate on the basis
  double s;
  double zz;              // |s z|^2
  double tau;
  double scale;           // reference scale max|new_outputs|
  double maxabs;
  uint32_t flags;
  int32_t skip;           // 1: proposal already complete (or eval mode): the residual pass skips it
};

// device-side copy of what the MH scan needs from a proposal's score
struct MhRes {
  double loglik;
  int32_t rank;
  int32_t pad;
};

struct ChainFitOut {
  double sse, scale;
  double beta[BSR_MAX_K + 1];      // weights on the scaled columns (reference's Beta before "/ scale")
  double beta_unscaled[BSR_MAX_K + 1];
  double maxabs[BSR_MAX_K];
  uint32_t colflags[BSR_MAX_K];
  uint32_t anyflags;
  uint32_t pad;
};

// ---- fast chain refresh (bsr_refresh.hip)
#define BSR_RF_ROWS 1024
struct RefreshIn {   // per chain, written by the host: what it already knows about the K current columns
  double colmax[BSR_MAX_K];
  uint32_t colflags[BSR_MAX_K];
};
struct RefreshPlan {  // per chain, device scratch handed from one refresh kernel to the next
  uint32_t anyflags, pad;
  double scale_fit;
  double beta_fit[BSR_MAX_K], coef_fit[BSR_MAX_K], beta_icpt[BSR_MAX_K + 1];
  double s_k[BSR_MAX_K], m_other[BSR_MAX_K];
  uint32_t flags_k[BSR_MAX_K];
  int32_t fallback, pad2;      // the Cholesky-QR factors are not accurate enough: rebuild with Gram-Schmidt
  double R1[BSR_NQ_MAX * BSR_NQ_MAX], T1[BSR_NQ_MAX * BSR_NQ_MAX], T2[BSR_NQ_MAX * BSR_NQ_MAX];
  double qy[BSR_NQ_MAX];
  double dcol[BSR_NQ_MAX];
};

// ---- tile-stationary row pass (bsr_tile.hip).  Fixed per context (so that a proposal's partial sums do not depend on
// the batch it is scored in): rows per lane and block, the number of row slices and their block ranges.
#ifndef BSR_TILE_WAVES
#define BSR_TILE_WAVES 16                 // waves per workgroup (one workgroup per CU)
#endif
#define BSR_TILE_U 2                      // rows per lane and block: a block is 64 * U = 128 rows
#define BSR_TILE_BLOCK (BSR_WAVE * BSR_TILE_U)
#define BSR_TILE_NB 4                     // blocks per pass of a chain tape (2 NB values per lane in registers)
#define BSR_TILE_ARG_GROUPS 4
#define BSR_TILE_ARG_COLS 32
#ifndef BSR_TILE_QMAX
#define BSR_TILE_QMAX 2
#endif                                    // tapes (sets of per-lane sums) per wave and pass: with four, the sums, a pass of
                                          // values and its operand columns no longer fit 128 registers (K = 3: 17 spilled)
struct TileGeom {
  int T;                // tape groups: workgroup w serves slice w % n_slices with the tapes of group w / n_slices
  int n_slices;         // row slices (partials per proposal)
  int bps;              // blocks per slice
  int n_blocks;         // ceil(N / BSR_TILE_BLOCK)
  int chunk_blocks;     // blocks staged in LDS at a time: bps (the whole slice, staged once), or fewer: the slice then
  int ring;             // streams through a ring of `ring` buffers (2..4) by LDS-DMA, ring - 1 chunks ahead
  int n_long;           // the first n_long slices hold bps + 1 blocks (streaming geometry: every block in a slice, n_left = 0)
  int n_left;           // blocks behind the last slice (n_blocks - n_slices * bps): (tape, block) units dealt to the waves
  int n_pass;           // passes over the slice (tapes per wave beyond the sets of sums)
  int qmax;             // sets of sums per wave of the launched kernel (tile_qmax(K))
  int ncols;            // most LDS columns of any tape group (X columns it reads, y, K basis columns per chain of the batch)
  int ncols_fixed;      // ... of which the basis columns: chains of the batch x K
  int n_part;           // partial records per proposal = n_slices + n_left
  int per_group;        // > 0: whole-slice variant with tape pulling (k_tile1): entries of a group's list of record indices
                        // (cost order, -1 padded), which sit behind the schedule's cost-order index
};
// Everything a wave needs to run one tape of the tile pass, in one 128-byte record the host writes into the schedule:
// the wave fetches it with one scalar load instead of three dependent ones (schedule -> descriptor -> streams) from
// memory that misses every cache -- the host wrote it microseconds ago.
struct TapeRec {
  int32_t p;            // tape (proposal) index, -1: no tape in this set of sums
  int32_t n_nodes;      // stream entries
  int32_t chain;        // chain tape (bsr_device.h: chain_eval)
  int32_t qslot;        // LDS slot of the chain's first basis column in the group's map
  double s;             // prescale of the candidate column
  uint64_t code0, code1, f0, f1;   // first words of the opcode and column streams
  double ln[6];         // first three (a, b) pairs of the ln stream
  int32_t code_off, feat_off, ln_off;   // where the streams go on
  int32_t n_ln, n_term;
  int32_t grp;          // the tape's group
};
static_assert(sizeof(TapeRec) == 128, "one tape record per 128 bytes");

// The streaming kernel's view of a tape (bsr_stream.hip): what its scalar-register interpreter needs, 32 bytes, one per
// (wave, set of sums) in schedule order and one of padding behind the last.
struct StreamRec {
  int32_t meta;         // bits 0..4: stream entries - 1, bits 5 and 31: fast (<= 16 entries, <= 8 terminals, <= 3 ln nodes,
                        // one value below the accumulator, no `log`), bit 6: there is a tape, bits 8..15: LDS slot of the
                        // chain's first basis column.  Round 6: bit 30 (without bits 5, 31) = the chunk block of assembly
                        // takes the tape and the C++ / tape-at-a-time interpreters do not (a program of several words, a
                        // second value below the accumulator at K <= 3) -- sorted out on the block's slow branch, so a 64-bit
                        // program costs what it did; bits 16..19: extension words behind this one, bits 20..29: the first
                        // one's place, in 16-byte units behind the wave's first StreamRec (the block's %[sr])
  int32_t first;        // LDS slot of the leading terminal x 1024
  double s;             // prescale of the candidate column
  uint64_t code;        // fast tapes: the entries behind the leading terminal, 4 bits each: operator + 1, 0 behind the last
  uint64_t slots;       // LDS slots of terminals 2..8, 8 bits each, in stream order
};
static_assert(sizeof(StreamRec) == 32, "one scalar load per tape");

// The whole-slice kernel's assembly tape loop (bsr_tile_asm.hip: k_tile1a) reads a tape as ONE 64-byte program, in the
// group's cost order (entry i of a group's array = entry i of its list; one record of padding, p = -1, behind the last).
struct TileProg {
  int32_t meta;         // bit 31: the block of assembly takes the tape (a chain of at most 17 entries, 8 terminals in LDS
                        // slots below 256, 2 ln nodes, no `log`); bits 0..7: LDS slot of the chain's first basis column
  int32_t p;            // tape (proposal) index, -1: padding
  double s;             // prescale of the candidate column
  uint64_t code;        // the entries behind the leading terminal, 4 bits each: operator + 1, 0 behind the last
  uint64_t slots;       // LDS slots of the terminals in stream order, 8 bits each (the leading terminal first)
  double ln[4];         // the (a, b) pairs of the first two ln nodes
};
static_assert(sizeof(TileProg) == 64, "one scalar load per tape");

template <typename T>
struct TileArgs {
  TileGeom g;
  const T* const* colsrc;     // [T][cols_stride] global column pointers (device memory): a tape group's X columns in LDS
  int cols_stride;            // slot order, its y, the basis columns of the batch's chains
  int cols_in_args;           // the groups' tables are also in `cols` below (T groups of arg_stride columns, 128 in all): the
                              // table in memory was written microseconds ago and misses every cache, the argument block
                              // is on its way to the workgroup anyway
  int arg_stride;             // columns per group of `cols` (32 for up to four groups; 64 for two, 128 for one)
  int grp_nF[8];              // X columns per tape group (= the LDS slot of its y)
  int64_t N;
  const uint64_t* codes;
  const uint64_t* feats;      // column stream in the tape group's LDS slots
  const double* lnp;
  const PropDesc* desc;
  const TapeRec* sched;       // [T][n_pass][BSR_TILE_WAVES][qmax] the waves' tapes (host: cost-balanced)
  const StreamRec* srec;      // the same schedule as StreamRecs (streaming kernel), else null
  const TileProg* tprog;      // whole-slice kernel with the assembly tape loop: [T][per_group + 1] programs, else null
  int split_stage;            // ... its staging in two halves (0: everything at the first barrier, as k_tile1 stages)
  double* part;               // [P][n_part][BSR_P1_WORDS]
  int P;
  int K;
  unsigned long long* stamps; // diagnostics (BSR_TILE_STAMPS=1): [workgroup][wave][8] clock samples, else null
  const T* cols[BSR_TILE_ARG_GROUPS][BSR_TILE_ARG_COLS];
};
#define BSR_TILE_STAMP_WORDS 8
#ifndef BSR_HOST_ONLY
template <typename T>
void launch_tile(hipStream_t st, const TileArgs<T>& a);
void launch_stream(hipStream_t st, const TileArgs<double>& a, bool deep2);   // bsr_stream.hip: fp64 slices that stream through LDS (deep2: the kernel with a second saved value)
void launch_stream_f32(hipStream_t st, const TileArgs<float>& a);   // ... f32 storage, f64 arithmetic (K <= 4)
void launch_tile_asm(hipStream_t st, const TileArgs<double>& a); // bsr_tile_asm.hip: whole slices, tape loop in assembly (a.tprog)
#endif
size_t tile_lds_bytes_max();
bool tile_asm_takes(int K);       // the assembly tape loop is written for these K (fp64, whole-slice contexts)
int tile_qmax(int K);
size_t stream_ln_bytes(int qt);   // LDS the streaming kernel needs behind its ring
int stream_qmax(int K);
bool stream_chunk_block(int K, int ncols_fixed);           // its sets of sums per wave
bool stream_deep2_applies(int K, int ncols_fixed, int chunk_blocks);
#define BSR_STREAM_LN_PAIRS 3     // (a, b) pairs of a tape the streaming kernel keeps in LDS (eight were tried in round 6 for
                                  // the long programs: 5 KB more LDS took the widest batches' ring from three buffers to two,
                                  // C5 74.5 -> 86 us per launch; 5 % of depth-12 trees hold more than three ln nodes and stay
                                  // with the stack machine)
#define BSR_STREAM_EXT_MAX 4      // extension words (16 entries, 8 terminal slots each) behind a tape's 64-bit program
#define BSR_STREAM_UNITS_MAX 64   // (column, block) pieces of a chunk the streaming kernel's waves can copy (4 per wave)

struct LaunchGeom {
  int rb_rows;      // rows per row block (multiple of 256)
  int n_rb;         // row blocks
  int pg;           // proposals per workgroup
  int n_pg;         // proposal groups
  int dyn_wgs;      // workgroups of the work-queue (projection) launch
};

// slot-explicit entry points behind bsr_score_submit / bsr_score_wait / bsr_commit (bsr_api.hip), used by the native
// sampler's worker threads: each thread owns one batch slot, and holds the context lock across commit + refresh + fit
__attribute__((visibility("hidden"))) int bsr_internal_submit(bsr_ctx* c, int slot, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                        const int32_t* which_k, const double* sigma, int32_t B);
__attribute__((visibility("hidden"))) int bsr_internal_wait(bsr_ctx* c, int slot, bsr_score* out);
__attribute__((visibility("hidden"))) int bsr_internal_submit_mh(bsr_ctx* c, int slot, const bsr_node* rows, const int32_t* tape_off,
                           const int32_t* chain, const int32_t* which_k, const double* sigma, int32_t B,
                           const double* terms8, const int32_t* flags, const int32_t* span_off, int32_t n_spans,
                           bool defer = false);
__attribute__((visibility("hidden"))) int bsr_internal_wait_mh(bsr_ctx* c, int slot, bsr_score* out, bsr_event* events);
__attribute__((visibility("hidden"))) int bsr_internal_commit(bsr_ctx* c, int slot, int32_t chain, int32_t k, int32_t idx);
__attribute__((visibility("hidden"))) void bsr_internal_feature_range(const bsr_ctx* c, const double** lo, const double** hi);
__attribute__((visibility("hidden"))) void bsr_internal_lock(bsr_ctx* c);
__attribute__((visibility("hidden"))) void bsr_internal_unlock(bsr_ctx* c);
// CPUs this process may use (cgroup quota / local ranks): sizes the submission threads and the sampler's worker threads
__attribute__((visibility("hidden"))) double bsr_internal_cpu_budget();
// CPUs of the L3 domain the library's threads are confined to (csrc/bsr_place.hip; 0: no placement)
__attribute__((visibility("hidden"))) int bsr_internal_placed_cpus();
// confines the calling thread (a thread the library started) to the library's CPUs: one L3 domain of the host
__attribute__((visibility("hidden"))) void bsr_internal_place_thread();

// kernels (bsr_kernels.hip)
// The finalise step fused behind the residual pass (K <= 3): its last workgroup to finish walks the flagged list
// (ck == null: not fused, k_finalize is launched instead).  The residual sums and `arrive` live in uncached memory.
struct FinArgs {
  const ChainB* ck;
  bsr_score* out;
  MhRes* mh;
  double rank_floor;
  int32_t* arrive;       // workgroup arrival counter (zero between launches)
};
template <typename T>
struct RowPassArgs {
  LaunchGeom g;
  const T* Xt;
  const T* y;
  int64_t ld, N;
  const uint64_t* codes;
  const uint64_t* feats;
  const double* lnp;
  const PropDesc* desc;
  const PropCoef* coef;
  int P;
  const int32_t* feat_list;   // non-null: stage these X columns (+y) in LDS; null: read X/y from global memory
  int nF;
  double* part;
  void* spill;
  int spill_slots;
  int rows_per_lane;          // 2, 4 or 8 (rb_rows must be a multiple of 64*rows_per_lane)
  int32_t* queue;             // work queue of the projection pass: 8 x BSR_QUEUE_SUB ticket counters, 128 B apart
  int32_t* queue_clear;       // the counter set this launch zeroes for a later one
  FinArgs fin;                // residual pass only
};
#ifndef BSR_HOST_ONLY
template <typename T>
void launch_rows(hipStream_t st, const RowPassArgs<T>& a, int nq, int residual);
// the finalise step can ride behind the residual pass for these K (register budget of its 16-wave workgroups)
static inline bool residual_can_fuse_finalize(int K) { return K <= 3; }
void launch_solve(hipStream_t st, const PropDesc* desc, const ChainB* ck, int P, int n_rb, const double* part1,
                  int64_t N, PropCoef* coef, bsr_score* out, double rank_floor, int32_t* flagged, MhRes* mh,
                  int32_t* flagged_next);
void launch_finalize(hipStream_t st, const PropDesc* desc, const ChainB* ck, const PropCoef* coef, int P, int n_rb,
                     const double* part2, int64_t N, bsr_score* out, double rank_floor, int32_t* flagged, MhRes* mh,
                     int n_wg);
void launch_events(hipStream_t st, const MhRes* mh, const double* terms8, const int32_t* flags, const int32_t* span_off,
                   int n_spans, int K, bsr_event* events);
template <typename T>
void launch_refresh_basis(hipStream_t st, const T* cur, T* Q, const T* y, int64_t ld, int64_t N, int K,
                          const double* col_maxabs, const uint32_t* col_flags, ChainB* cb);
template <typename T>
void launch_chain_fit(hipStream_t st, const T* cols, const T* y, int64_t ld, int64_t N, int K, int intercept,
                      ChainFitOut* out);
template <typename T>
void launch_transpose_in(hipStream_t st, const double* src_rowmajor, T* dst, int64_t N, int d, int64_t ld);
template <typename T>
void launch_convert_out(hipStream_t st, const T* src, double* dst, int64_t n);
template <typename T>
void launch_convert_in(hipStream_t st, const double* src, T* dst, int64_t n);
template <typename T>
void launch_refresh_fast(hipStream_t st, const T* cols, T* Q, const T* y, int64_t ld, int64_t N, int K,
                         const RefreshIn* d_in, RefreshPlan* d_plan, double* d_part, ChainB* ck,
                         ChainFitOut* fit_noicpt, ChainFitOut* fit_icpt);
size_t refresh_part_doubles(int64_t N);
#endif  // BSR_HOST_ONLY
