// C ABI of libbsr_hip.so (include/bsr_hip.h): context, chain cache, batch scoring, RCCL gather.
// Host-side orchestration only; all O(N) work is in bsr_kernels.hip.
#include <rccl/rccl.h>

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "bsr_ctx.h"

thread_local std::string g_create_error;
std::atomic<long long> g_ns_stage{0}, g_ns_desc{0}, g_ns_enq{0}, g_n_sub{0};
std::atomic<long long> g_ns_issue{0}, g_n_issue{0}, g_ns_wait{0}, g_n_wait{0};
std::atomic<long long> g_ns_part[12];   // issue_batch, piece by piece: host staging, upload, row pass, solve, residual pass, finalize + events, done event
const bool g_host_prof = getenv("BSR_HOST_PROF") != nullptr;
const bool g_stream_stats = getenv("BSR_STREAM_STATS") != nullptr;
std::atomic<long long> g_ss_entries[2][16], g_ss_tapes[2], g_ss_batches{0}, g_ss_derived{0}, g_ss_cols{0};

static void launcher_start(bsr_ctx* c);
static void launcher_stop(bsr_ctx* c);
static int build_derived(bsr_ctx* c);

void set_err(bsr_ctx* c, const char* msg) {
  std::lock_guard<std::mutex> lk(c->err_mu);
  c->err = msg;
}

int fail(bsr_ctx* ctx, int code, const char* msg) {
  if (ctx) set_err(ctx, msg); else g_create_error = msg;
  return code;
}


extern "C" int bsr_abi_version(void) { return BSR_ABI_VERSION; }

extern "C" int bsr_device_count(int* count) {
  if (!count) return BSR_E_ARG;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return BSR_E_NODEVICE;
  }
  *count = n;
  return BSR_OK;
}

extern "C" const char* bsr_last_error(const bsr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// BSR_POISON=1 (debugging): device buffers that are handed out without a fill are filled with 0xFF bytes (NaN doubles,
// -1 integers) -- a kernel that reads what nobody wrote shows at once instead of depending on what the allocator
// happened to return.
void poison(void* p, size_t bytes) {
  // (read at every allocation, not once per process: tests/test_gpu_ctx_sequence.py interleaves poisoned and clean lives
  // of the same blocks)
  const char* v = getenv("BSR_POISON");
  const bool on = v && atoi(v) != 0;
  if (on && p && bytes) { (void)hipMemset(p, 0xFF, bytes); (void)hipDeviceSynchronize(); }
}
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

static int wait_slot(bsr_ctx* c, BatchSlot& s);

extern "C" int bsr_ctx_destroy(bsr_ctx* c) {
  if (g_host_prof && g_n_sub.load() > 0) {
    const double n = (double)g_n_sub.load();
    fprintf(stderr, "bsr host cost per submission: stage %.2f us, descriptors %.2f us, enqueue (sort + hand-over or HIP calls) %.2f us over %.0f; "
            "issuing a batch's HIP calls %.2f us (%.0f batches); wait %.2f us (%.0f waits)\n",
            g_ns_stage.load() / n * 1e-3, g_ns_desc.load() / n * 1e-3, g_ns_enq.load() / n * 1e-3, n,
            g_ns_issue.load() / std::max(1.0, (double)g_n_issue.load()) * 1e-3, (double)g_n_issue.load(),
            g_ns_wait.load() / std::max(1.0, (double)g_n_wait.load()) * 1e-3, (double)g_n_wait.load());
    const double ni = std::max(1.0, (double)g_n_issue.load());
    fprintf(stderr, "  issuing, piece by piece (us): staging on this thread %.2f, upload %.2f, row pass launch %.2f, k_solve launch %.2f, "
            "residual pass launch %.2f, finalize / events launches %.2f, done event %.2f\n",
            g_ns_part[0].load() / ni * 1e-3, g_ns_part[1].load() / ni * 1e-3, g_ns_part[2].load() / ni * 1e-3,
            g_ns_part[3].load() / ni * 1e-3, g_ns_part[4].load() / ni * 1e-3, g_ns_part[5].load() / ni * 1e-3,
            g_ns_part[6].load() / ni * 1e-3);
    fprintf(stderr, "  staging on the issuing thread (us): mark_in_span %.2f, stage_tile %.2f, build_tile_launch %.2f\n",
            g_ns_part[7].load() / ni * 1e-3, g_ns_part[8].load() / ni * 1e-3, g_ns_part[9].load() / ni * 1e-3);
  }
  if (g_stream_stats && g_ss_batches.load() > 0) {
    static const char* names[16] = {"inv", "ln", "neg", "sin", "cos", "exp", "square", "cubic", "add", "mul", "T", "add_T",
                                    "mul_T", "sub", "div", "log"};
    const double nb = (double)g_ss_batches.exchange(0);
    fprintf(stderr, "bsr stream stats over %.0f batches: per batch %.2f chain tapes, %.2f stack tapes, %.2f derived columns, %.2f X columns\n",
            nb, g_ss_tapes[0].load() / nb, g_ss_tapes[1].load() / nb, g_ss_derived.load() / nb, g_ss_cols.load() / nb);
    for (int kind = 0; kind < 2; ++kind) {
      fprintf(stderr, "  %s entries per batch:", kind ? "stack" : "chain");
      for (int o = 0; o < 16; ++o) fprintf(stderr, " %s %.2f", names[o], g_ss_entries[kind][o].exchange(0) / nb);
      fprintf(stderr, "\n");
      g_ss_tapes[kind].store(0);
    }
    g_ss_derived.store(0);
    g_ss_cols.store(0);
  }
  if (!c) return BSR_E_ARG;
  (void)hipSetDevice(c->device);
  launcher_stop(c);
  if (c->poisoned.load()) {   // a batch that never completed may still run: nothing it can reach is freed or destroyed
    delete c;
    return BSR_OK;
  }
  for (BatchSlot& s : c->slot)   // (a batch nobody waited for: its kernels, dispatched directly, are on no stream)
    if (s.pending) (void)wait_slot(c, s);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
  for (BatchSlot& s : c->slot) {
    if (s.stream) (void)hipStreamSynchronize(s.stream);
  }
  void* dev[] = {c->Xt, c->y, c->cur, c->Q, c->zbuf, c->d_ck, c->d_fit, c->d_stage, c->comm_buf, c->d_fit_icpt,
                 c->d_rin, c->d_plan, c->d_rpart, c->d_stamps};
  if (c->h_plan) (void)hipHostFree(c->h_plan);
  for (void* p : dev) if (p) (void)hipFree(p);
  for (BatchSlot& s : c->slot) {
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.d_coef) (void)hipFree(s.d_coef);
    if (s.d_flagged) (void)hipFree(s.d_flagged);
    if (s.d_mh) (void)hipFree(s.d_mh);
    if (s.h_ev) (void)hipHostFree(s.h_ev);
    if (s.queue) (void)hipFree(s.queue);
    if (s.part1) (void)hipFree(s.part1);
    if (s.part2) (void)hipFree(s.part2);
    if (s.spill) (void)hipFree(s.spill);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.h_done_word) (void)hipHostFree(s.h_done_word);
    if (c->aql) aql_slot_destroy(c->aql, &s.aql);
    for (auto& e : s.ev) if (e) (void)hipEventDestroy(e);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return BSR_OK;
}

template <typename T>
static int upload_data(bsr_ctx* c, const double* X, const double* y) {
  // stage the row-major host matrix on the device, then transpose to feature-major [d][ld]
  double* stage = nullptr;
  const size_t xbytes = (size_t)c->N * c->d * sizeof(double);
  HIPCHK(c, hipMalloc((void**)&stage, std::max(xbytes, (size_t)c->N * sizeof(double))));
  HIPCHK(c, hipMemcpyAsync(stage, X, xbytes, hipMemcpyHostToDevice, c->stream));
  launch_transpose_in<T>(c->stream, stage, (T*)c->Xt, c->N, c->d, c->ld);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (y) {
    HIPCHK(c, hipMemcpyAsync(stage, y, (size_t)c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_convert_in<T>(c->stream, stage, (T*)c->y, c->N);
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  HIPCHK(c, hipFree(stage));
  return BSR_OK;
}

extern "C" int bsr_ctx_create(bsr_ctx** out, int device, int64_t N, int32_t d, const double* X, const double* y,
                              int32_t K, int32_t n_chains, int32_t max_batch, int32_t dtype) {
  return bsr_ctx_create_tuned(out, device, N, d, X, y, K, n_chains, max_batch, dtype, 0, 0);
}

extern "C" int bsr_ctx_create_tuned(bsr_ctx** out, int device, int64_t N, int32_t d, const double* X, const double* y,
                                    int32_t K, int32_t n_chains, int32_t max_batch, int32_t dtype, int32_t typical_chains,
                                    int32_t typical_batch) {
  if (!out) return BSR_E_ARG;
  *out = nullptr;
  if (!X || N <= 0 || d <= 0 || d > 65536) return fail(nullptr, BSR_E_ARG, "bsr_ctx_create: bad X/N/d");
  if (K < 0 || K > BSR_MAX_K || n_chains < 0 || max_batch <= 0)
    return fail(nullptr, BSR_E_ARG, "bsr_ctx_create: bad K/n_chains/max_batch");
  if (typical_chains < 0 || typical_batch < 0) return fail(nullptr, BSR_E_ARG, "bsr_ctx_create_tuned: bad typical_chains/typical_batch");
  if (dtype != BSR_DTYPE_F64 && dtype != BSR_DTYPE_F32) return fail(nullptr, BSR_E_ARG, "bsr_ctx_create: bad dtype");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
    return fail(nullptr, BSR_E_NODEVICE, "bsr_ctx_create: no such HIP device");
  choose_lib_cpus(device);   // (once per process: an L3 domain of this device's NUMA node)
  bsr_ctx* c = new bsr_ctx();
  c->device = device;
  c->N = N;
  c->ld = (N + BSR_ROW_ALIGN - 1) / BSR_ROW_ALIGN * BSR_ROW_ALIGN;
  c->d = d;
  c->K = K;
  c->n_chains = n_chains;
  c->max_batch = max_batch;
  c->dtype = dtype;
  c->esz = (dtype == BSR_DTYPE_F64) ? 8 : 4;
  c->has_y = (y != nullptr);
#ifdef BSR_TEST_VARIANTS
  c->rows_per_lane = env_int("BSR_P1_U", 2);
#else
  c->rows_per_lane = 2;   // (4 and 8 rows per lane and sweep: built into the test library only)
#endif
  if (c->rows_per_lane != 2 && c->rows_per_lane != 4 && c->rows_per_lane != 8) c->rows_per_lane = 2;
  // rows per task: 1024 amortises the per-task setup and reductions (~1 us) once there are enough row blocks to go
  // round (measured: +5..13 % at N = 100k..1M); small inputs keep shorter blocks so fewer masked rows are evaluated
  const int rb_default = (N >= 65536) ? 1024 : ((N > 256) ? 512 : 256);
  c->rb_rows = env_int("BSR_RB_ROWS", rb_default);
  if (c->rb_rows < 256 || c->rb_rows > BSR_ROW_ALIGN || (BSR_ROW_ALIGN % c->rb_rows) != 0 ||
      (c->rb_rows % (64 * c->rows_per_lane)) != 0)
    c->rb_rows = rb_default;
  if (c->rb_rows % (64 * c->rows_per_lane) != 0) c->rows_per_lane = 2;  // a row block is a whole number of sweeps
#ifdef BSR_TEST_VARIANTS
  c->no_lds = env_int("BSR_NO_LDS", 1);  // measured: at the headline workload reading X from L2 beats LDS staging
#else
  c->no_lds = 1;                         // (k_rows with X staged in LDS, round 1's static grid: test library only)
#endif
  // derived columns (see kDerivedOps): on unless asked off, the column ids would leave 16 bits, or X is so large that
  // nine copies of it would take more than a third of the device's free memory
  c->n_cols = d;
  if (env_int("BSR_DERIVED", 1) && (int64_t)d * (1 + kNumDerivedOps) <= 65535) {
    size_t free_b = 0, total_b = 0;
    (void)hipSetDevice(device);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
        (double)(BSR_ROW_ALIGN + N) * c->esz * d * (1 + kNumDerivedOps) <= (double)free_b / 3.0)
      c->n_cols = d * (1 + kNumDerivedOps);
  }
  for (BatchSlot& s : c->slot) s.slot_of.assign(c->n_cols, -1);
  int rc = BSR_OK;
  auto bail = [&](int code) {
    g_create_error = c->err;
    bsr_ctx_destroy(c);
    return code;
  };
#define CK(call)                                       \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) {                            \
      c->err = std::string(#call) + ": " + hipGetErrorString(e_); \
      return bail(BSR_E_HIP);                          \
    }                                                  \
  } while (0)
  CK(hipSetDevice(device));
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
    // device memory mapped into the host's address space (large BAR): input blocks are written there directly
    // (issue_batch; tools/probes/bar_write_probe.hip).  BSR_BAR_WRITE=0: hipMemcpyAsync as before.
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess) large_bar = 0;
    c->bar_write = (large_bar && env_int("BSR_BAR_WRITE", 1)) ? 1 : 0;
  }
  {
    // Tile pass: one workgroup per CU.  T tape groups share the CUs: n_cu / T row slices, each served by T
    // workgroups with different tapes.  More groups = fewer tapes per wave and longer slices (fewer lane reductions,
    // less padding at the slice ends), but every group stages the slice's columns again: only while the data set is
    // L2-sized.  Everything here depends on the context alone.
    c->tile_on = env_int("BSR_TILE", 1);

    c->selfdup = env_int("BSR_SELFDUP", 1);
    c->solve_exact = env_int("BSR_SOLVE_EXACT", 0);
    c->stream_deep = env_int("BSR_STREAM_DEEP", 2);
    c->reorder = env_int("BSR_REORDER", 1);
    c->chain_eval = env_int("BSR_CHAIN_EVAL", 1);
    // more than a handful of derived columns per batch stop paying: most (op, feature) pairs of a batch are used by one
    // tape, and staging a column costs every workgroup what one use saves one wave (measured flat from 7 up)
    c->derived_max = env_int("BSR_DERIVED_MAX", 8);
    // Launch width of the tile pass: n_cu - BSR_AUX_CUS workgroups, by default three quarters of the CUs.  A tile
    // workgroup needs a whole CU (LDS and registers), and a launch as wide as the machine ends when its last workgroup
    // does: the CUs that finish early wait, the next batch's launch starts staging only then, and a CU that holds
    // another batch's solve or residual waves holds a workgroup up.  With narrower launches the next batch's workgroups
    // take the CUs as they come free -- staging of one batch runs under the arithmetic of another, and the small
    // kernels find room.  Each workgroup then has a third more rows (8 blocks per slice instead of 6 at N = 100k), the
    // kernel alone takes 25 instead of 21.5 us, and the pipelined step 18.9 instead of 21.1 us at K = 3 (29.0 instead
    // of 32.8 at K = 8); half the CUs measures the same at K = 3 and worse at K = 8, a quarter loses.
    // (tools/probes/aux_cus_sweep.sh)
    c->aux_cus = env_int("BSR_AUX_CUS", c->n_cu / 4);
    if (c->aux_cus < 0 || c->aux_cus > c->n_cu * 7 / 8) c->aux_cus = 0;
    // Geometry of the row pass (bsr_tile.hip), fixed for the life of the context.  A wave holds tile_qmax(K) sets of
    // sums, so one group of 16 waves takes 16 x qmax tapes per pass over its slice; `want` groups give every tape of the
    // widest batch a set.  Slices that fit LDS whole are staged once (the largest T <= want for which they do even for
    // the widest batch: every X column, y, every chain's basis); otherwise the slice streams through two LDS buffers
    // chunk by chunk, every CU takes part (no spare CUs: the pass is then bound by HBM), and T = 2 groups by default so
    // that a group's columns -- its tapes' features, y, the basis -- leave room for chunks of a few blocks.
    c->tile_qmax = tile_qmax(std::max(1, K));
    c->tile_blocks = (int)((N + BSR_TILE_BLOCK - 1) / BSR_TILE_BLOCK);
    // The batch the geometry is chosen for: the context's limits, or what the caller says its batches look like
    // (bsr_ctx_create_tuned: the native sampler's chain groups submit a quarter of the chains at a time).  A wider
    // batch still scores, to the same bytes, over the same slices -- through the chunked kernel where its columns no
    // longer fit LDS whole.
    const int geo_chains = std::max(1, typical_chains > 0 ? std::min((int)n_chains, (int)typical_chains) : (int)n_chains);
    const int geo_batch = std::max(1, typical_batch > 0 ? std::min((int)max_batch, (int)typical_batch) : (int)max_batch);
    const int want = std::max(1, std::min(8, (geo_batch + BSR_TILE_WAVES * c->tile_qmax - 1) / (BSR_TILE_WAVES * c->tile_qmax)));
    const size_t worst_cols = (size_t)d + 1 + (size_t)geo_chains * std::max(1, K);
    const size_t budget = tile_lds_bytes_max() - 1024;
    for (int attempt = 0; attempt < 2; ++attempt) {
      c->tile_cus = c->n_cu - c->aux_cus;
      // Batches of at most four tapes per wave of ONE workgroup (64): a single tape group over slices of eight blocks
      // -- what a chain tape evaluates in one pass -- where such a slice fits LDS whole for the widest batch.  Half as
      // many workgroups as two groups over the same slices, each staging the slice once for all tapes and dealing four
      // tapes per wave instead of two (C2: 96 workgroups, 26.5 us alone instead of 192 at 19.7, but 2 600 instead of
      // 3 900 CU-us per batch: 11.7-11.9 instead of 12.7-13.3 us per pipelined step).  A launch narrower than the
      // machine leaves the other CUs to the batches behind it.
      const int long_bps = std::max(2, std::min(16, env_int("BSR_TILE_BPS", 8)));   // (6, 5, 4 and 10 blocks per slice measured 1.2-2.4 us per step slower, interleaved in one box)
      // slices of WHOLE blocks only (the block that holds row N, N not a multiple of 128, is a leftover unit -- the one place
      // that masks rows): `whole` blocks dealt over the slices, the first `long` of them one block longer -- so a
      // slice count that leaves no slice above long_bps blocks
      const int whole = (int)(N / BSR_TILE_BLOCK);
      const int long_slices = (whole + long_bps - 1) / long_bps;
      // (from 32 slices on: a short data set keeps the many short slices -- one batch at a time is what it is scored in)
      const bool long_ok = geo_batch <= 4 * BSR_TILE_WAVES && long_slices >= 32 && long_slices <= c->n_cu &&
                           worst_cols * (size_t)long_bps * BSR_TILE_BLOCK * c->esz <= budget && !getenv("BSR_TILE_T") &&
                           env_int("BSR_TILE_LONG", 1);
      if (long_ok) c->tile_cus = long_slices;
      // Batches wider than that, of several chains (config 4's eight chains x 32 proposals in one launch): tape groups BY
      // CHAIN over the same long slices.  A group then stages its own chains' basis columns only -- two chains of eight:
      // 17 columns instead of 35 -- so the long slice fits LDS whole for every group, where every chain's columns
      // forced 256 slices of three blocks (half a workgroup's time was its launch ramp, first staging round and
      // reduction).  T x long_slices workgroups: more than CUs, they take them as they come free.
      int by_chain_T = 0;
      size_t group_cols = worst_cols;
      if (!long_ok && c->dtype == BSR_DTYPE_F64 && geo_chains >= 2 && long_slices >= 32 && long_slices <= c->n_cu &&
          !getenv("BSR_TILE_T") && env_int("BSR_TILE_LONG", 1) && env_int("BSR_TILE_BY_CHAIN", 1)) {
        for (int t = 2; t <= 8 && by_chain_T == 0; t *= 2) {
          const size_t cols_t = (size_t)d + 1 + (size_t)((geo_chains + t - 1) / t) * std::max(1, K);
          if (t <= geo_chains && (geo_batch + t - 1) / t <= 4 * BSR_TILE_WAVES &&
              cols_t * (size_t)long_bps * BSR_TILE_BLOCK * c->esz <= budget) {
            by_chain_T = t;
            group_cols = cols_t;
          }
        }
      }
      c->tile_by_chain = by_chain_T > 0;
      if (by_chain_T > 0) c->tile_cus = long_slices * by_chain_T;
      // blocks of the longest slice when `whole` blocks are dealt over tile_cus / t slices
      auto slice_blocks = [&](int t) { const int sl = std::max(1, c->tile_cus / t); return std::max(1, (whole + sl - 1) / sl); };
      auto fits_whole = [&](int t) { return group_cols * (size_t)slice_blocks(t) * BSR_TILE_BLOCK * c->esz <= budget; };
      // slices that fit LDS whole (even for the widest batch) are staged once and the waves pull their tapes (k_tile1):
      // the largest T in {4, 2, 1} with a tape per wave at most -- fewer tapes per wave and longer slices (fewer lane
      // reductions, record fetches and decodes per row), while every group stages the slice's columns again
      int T = 0;
      if (by_chain_T > 0 && fits_whole(by_chain_T)) T = by_chain_T;
      else c->tile_by_chain = false;
      if (T == 0 && long_ok && fits_whole(1)) T = 1;
      for (int t = 4; t >= 1 && T == 0; t >>= 1)
        if (t <= (geo_batch + BSR_TILE_WAVES - 1) / BSR_TILE_WAVES && c->tile_cus % t == 0 && fits_whole(t)) T = t;
      if (T == 0 && fits_whole(1)) T = 1;
      c->tile_whole = T > 0;
      if (!c->tile_whole) {
        if (c->aux_cus != 0) {   // chunked: all CUs
          c->aux_cus = 0;
          continue;
        }
        // fp64: the streaming kernel (bsr_stream.hip) -- four sets of sums per wave at K <= 4, so ONE tape group takes a
        // batch of 64 and every column streams from HBM once per launch; BSR_STREAM=0: the chunked k_tile as before
        // fp32 columns (round 6): the same kernel with f32 STORAGE -- half the bytes through HBM and the ring, every value
        // converted where it is read, the interpreter and the sums the f64 ones (K <= 4: the chunk block of assembly;
        // BSR_STREAM_F32=0: the chunked k_tile<float> with f32 tree arithmetic, as until round 5)
        c->tile_stream = env_int("BSR_STREAM", 1) != 0 &&
                         (c->dtype == BSR_DTYPE_F64 || (K <= 4 && env_int("BSR_STREAM_F32", 1) != 0));
        if (c->tile_stream) {
          c->tile_qmax = stream_qmax(std::max(1, K));
          T = 1;
          while (T < 8 && T * BSR_TILE_WAVES * c->tile_qmax < max_batch) T *= 2;
        } else {
          T = want;
          while (T > 4) T = (T + 1) / 2;
        }
      }
      T = std::max(1, std::min(8, env_int("BSR_TILE_T", T)));
      while (c->tile_whole && T > 1 && (c->tile_cus % T) != 0) --T;   // (chunked: n_cu / T slices, a CU or two may idle)
      if (c->tile_whole && !fits_whole(T)) c->tile_whole = false;
      c->tile_T = T;
      c->tile_slices = std::max(1, c->tile_cus / T);
      // every WHOLE block in a slice, the first tile_long slices one block longer (round 5: whole-slice contexts too --
      // before, what n_blocks mod n_slices left behind the last slice went out as (tape, block) units through L2, 6 x 64
      // of them at C2, and a data set whose block count was a multiple of the slice count while N was not a multiple of
      // 128 lost a block per slice to them); the block that holds row N (N not a multiple of 128) goes out as (tape,
      // block) units behind the loop: nothing in the loop masks rows
      c->tile_bps = whole / c->tile_slices;
      c->tile_long = whole - c->tile_bps * c->tile_slices;
      c->tile_left = c->tile_blocks - whole;
      // ... where most slices are long ones (C2: 95 of 98 slices hold 8 blocks, 3 hold 7).  Where few would be (C3: 13 of
      // 192 slices with a fifth block: the launch ends with its longest workgroups, +25 % for a sixteenth of them --
      // measured 24.8 against 22.5 us), the blocks behind bps x slices go out as (tape, block) units as before
      if (!c->tile_stream && c->tile_long * 2 < c->tile_slices) {
        c->tile_left += c->tile_long;
        c->tile_long = 0;
      }
      break;
    }
    c->tile_asm = c->tile_whole && c->dtype == BSR_DTYPE_F64 && tile_asm_takes(K) && env_int("BSR_TILE_ASM", 1) != 0;
    c->tile_split = env_int("BSR_TILE_SPLIT", 1) != 0;
    c->done_word = env_int("BSR_DONE_WORD", 1) != 0;
    c->tile_sched_cap = (size_t)c->tile_T * BSR_TILE_WAVES * c->tile_qmax *
                        (size_t)((max_batch + BSR_TILE_WAVES * c->tile_qmax - 1) / (BSR_TILE_WAVES * c->tile_qmax) + 1) + 64;
    // chunked: the transcendentals saved are worth more columns (C5, BSR_DERIVED_MAX 0 / 4 / 8 / 16 / 24 / 32: 95.6 / 82.7 / 80.8 /
    // 74.6-77.2 / 76.2 / 77.4 us per launch in f64 -- the kernel is bound by instruction issue, not by HBM: a column read
    // instead of recomputed pays even at 1.37 x the algorithmic traffic; f32 storage, half the bytes and LDS per column:
    // 8 / 16 / 24 / 32: 83.9 / 78.0 / 74.1 / 73.3 us, profiles/r06_derived_sweep_c5.txt)
    if (!c->tile_whole && !getenv("BSR_DERIVED_MAX")) c->derived_max = (c->esz == 4 && c->tile_stream) ? 24 : 16;
    // a data set of which not even one block of a narrow group (eight features or all of them, y, one chain's basis)
    // fits two LDS buffers never takes the tile pass: the work-queue row pass (bsr_kernels.hip: k_rows) serves it
    c->tile_ever = c->tile_on && (size_t)(std::min(d, 8) + 1 + std::max(1, K)) * 2 * BSR_TILE_BLOCK * c->esz <= budget;
    // The chunked kernel is built for data sets much larger than LDS (N = 1M: 30 blocks per CU).  A short data set whose
    // slices miss LDS only because its widest batch would carry many chains' bases (8 chains at K = 8: 75 columns at
    // N = 100k -- 3 blocks per CU) is better served by the work-queue pass: the native sampler's batches there (two
    // chains, 32 proposals each) measured 0.66-0.69 M consumed proposals/s through it against 0.55 M through k_tile.
    if (!c->tile_whole && c->tile_blocks < 8 * c->n_cu) c->tile_ever = false;
    // derived columns pay where the slice sits in LDS whole (a derived column is then one more column staged from L2);
    // a chunked pass would stream each of them from HBM (N = 1M: 8 MB per column and group), the work-queue pass for
    // every tape again
    if (!c->tile_ever && c->n_cols > d && env_int("BSR_DERIVED", 1) < 2) {
      c->n_cols = d;
      for (BatchSlot& s : c->slot) s.slot_of.assign(c->n_cols, -1);
    }
    if (env_int("BSR_TILE_STAMPS", 0) > 0) {
      // BSR_TILE_STAMPS=R: a ring of R launches' stamps (launch n writes block n mod R; R = 1: the last launch's)
      c->stamp_ring = std::min(4096, env_int("BSR_TILE_STAMPS", 0));
      c->stamp_block_words = (size_t)std::max(c->n_cu, c->tile_cus) * BSR_TILE_WAVES * BSR_TILE_STAMP_WORDS;
      const size_t nb = c->stamp_block_words * (size_t)c->stamp_ring * sizeof(unsigned long long);
      if (hipMalloc((void**)&c->d_stamps, nb) == hipSuccess) (void)hipMemset(c->d_stamps, 0, nb);
      else { c->d_stamps = nullptr; c->stamp_ring = 0; }
    }
  }
  CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  const size_t colb = (size_t)c->ld * c->esz;
  CK(hipMalloc(&c->Xt, colb * c->n_cols));
  CK(hipMemsetAsync(c->Xt, 0, colb * c->n_cols, c->stream));
  CK(hipMalloc(&c->y, colb));
  CK(hipMemsetAsync(c->y, 0, colb, c->stream));
  if (K > 0 && n_chains > 0) {
    CK(hipMalloc(&c->cur, colb * n_chains * K));
    CK(hipMemsetAsync(c->cur, 0, colb * n_chains * K, c->stream));
    CK(hipMalloc(&c->Q, colb * n_chains * K));
    CK(hipMemsetAsync(c->Q, 0, colb * n_chains * K, c->stream));
    CK(hipMalloc((void**)&c->d_ck, sizeof(ChainB) * n_chains));
  poison(c->d_ck, sizeof(ChainB) * n_chains);
    CK(hipMemsetAsync(c->d_ck, 0, sizeof(ChainB) * n_chains, c->stream));
    c->h_ck.resize((size_t)n_chains);
    c->ready.assign(n_chains, 0);
    c->col_set.assign((size_t)n_chains * K, 0);
    c->cur_form.assign((size_t)n_chains * K, bsr_span::LinForm());
    c->cur_form_ok.assign((size_t)n_chains * K, 0);
    c->span.assign((size_t)n_chains, std::shared_ptr<const bsr_span::SpanBasis>());
  }
  CK(hipMalloc((void**)&c->d_fit, sizeof(ChainFitOut) * (n_chains + 1)));
  poison(c->d_fit, sizeof(ChainFitOut) * (n_chains + 1));
  c->h_fit.resize(n_chains + 1);
  CK(hipMalloc((void**)&c->d_fit_icpt, sizeof(ChainFitOut) * (n_chains + 1)));
  poison(c->d_fit_icpt, sizeof(ChainFitOut) * (n_chains + 1));
  c->h_fit_icpt.resize(n_chains + 1);
  CK(hipMalloc((void**)&c->d_rin, sizeof(RefreshIn) * (n_chains + 1)));
  poison(c->d_rin, sizeof(RefreshIn) * (n_chains + 1));
  c->h_rin.resize(n_chains + 1);
  CK(hipMalloc((void**)&c->d_plan, sizeof(RefreshPlan)));
  poison(c->d_plan, sizeof(RefreshPlan));
  CK(hipHostMalloc((void**)&c->h_plan, sizeof(RefreshPlan)));
  CK(hipMalloc((void**)&c->d_rpart, refresh_part_doubles(N) * sizeof(double)));
  poison(c->d_rpart, refresh_part_doubles(N) * sizeof(double));
  c->fast_refresh = env_int("BSR_FAST_REFRESH", 1);
  for (BatchSlot& s : c->slot) {
    CK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    CK(hipMalloc((void**)&s.d_coef, sizeof(PropCoef) * (max_batch + 1)));
    poison(s.d_coef, sizeof(PropCoef) * (max_batch + 1));
    s.flag_stride = (size_t)max_batch + 2;
    CK(hipMalloc((void**)&s.d_flagged, sizeof(int32_t) * (2 * (max_batch + 2) + 16)));   // two lists, alternating by batch; then k_finalize's arrival counter
    CK(hipMemset(s.d_flagged, 0, sizeof(int32_t) * (2 * (max_batch + 2) + 16)));
    CK(hipMalloc((void**)&s.d_mh, sizeof(MhRes) * (max_batch + 1)));
    poison(s.d_mh, sizeof(MhRes) * (max_batch + 1));
    CK(hipMalloc((void**)&s.queue, (size_t)BSR_QUEUE_SETS * BSR_QUEUE_SET_INTS * sizeof(int32_t)));
    CK(hipMemset(s.queue, 0, (size_t)BSR_QUEUE_SETS * BSR_QUEUE_SET_INTS * sizeof(int32_t)));
    s.off_cols = ((size_t)c->n_cols * sizeof(int32_t) + 255) / 256 * 256;
    s.cols_stride = c->n_cols + 1 + std::max(1, n_chains) * std::max(1, K);   // column-pointer table: one per tape group
    s.off_sched = s.off_cols + ((size_t)c->tile_T * s.cols_stride * sizeof(void*) + 255) / 256 * 256;
    s.mh_cap = (size_t)max_batch;
    s.off_mh = s.off_sched;   // (the tile schedule lives behind the batch's streams: only what a batch uses is uploaded)
    s.off_desc = s.off_mh + (s.mh_cap * 8 * sizeof(double) + (2 * s.mh_cap + 2) * sizeof(int32_t) + 255) / 256 * 256;
    CK(hipHostMalloc((void**)&s.h_ev, sizeof(bsr_event) * (max_batch + 1)));
    s.off_streams = s.off_desc + sizeof(PropDesc) * (max_batch + 1);
    s.chain_slot.assign(std::max(1, n_chains), -1);
    CK(hipHostMalloc((void**)&s.h_out, sizeof(bsr_score) * (max_batch + 1)));
    CK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    if (hipHostMalloc((void**)&s.h_done_word, 64) == hipSuccess) memset(s.h_done_word, 0, 64);
    else { (void)hipGetLastError(); s.h_done_word = nullptr; }
    for (auto& e : s.ev) CK(hipEventCreate(&e));
  }
  if (c->bar_write) {
    // direct dispatch of the batches' kernels; a refusal (BSR_AQL=0, ROCr says no) leaves the streams
    const char* why = nullptr;
    c->aql = aql_device(device, &why);
    if (!c->aql && env_int("BSR_AQL_VERBOSE", 0)) fprintf(stderr, "bsr: no direct dispatch (%s)\n", why ? why : "?");
    if (c->aql) {
      int idx = 0;
      for (BatchSlot& s : c->slot) {
        s.aqb.reset(new AqlBatch());
        if (aql_slot_init(c->aql, &s.aql, idx++) != 0) c->aql_off = true;
      }
    }
  }
#undef CK
  c->x_lo.assign(d, INFINITY);
  c->x_hi.assign(d, -INFINITY);
  for (int64_t n = 0; n < N; ++n) {
    const double* row = X + (size_t)n * d;
    for (int f = 0; f < d; ++f) {
      c->x_lo[f] = std::min(c->x_lo[f], row[f]);   // NaN entries leave the range alone
      c->x_hi[f] = std::max(c->x_hi[f], row[f]);
    }
  }
  rc = (dtype == BSR_DTYPE_F64) ? upload_data<double>(c, X, y) : upload_data<float>(c, X, y);
  if (rc != BSR_OK) return bail(rc);
  // every buffer initialised above (null-stream and main-stream memsets) is complete before any slot stream runs
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(c, BSR_E_HIP, "hipDeviceSynchronize after setup"));
  if (c->n_cols > d) {
    rc = build_derived(c);
    if (rc != BSR_OK) return bail(rc);
  }
  if (env_int("BSR_SUBMIT_THREAD", 1)) launcher_start(c);
  *out = c;
  return BSR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// A chain's current tree k changed (set_current / commit): its linear form and the chain's span basis (bsr_span.h).
static void note_current_tree(bsr_ctx* c, int chain, int k, const bsr_node* t, int len) {
  const size_t ck = (size_t)chain * c->K + k;
  c->cur_form_ok[ck] = (t && len > 0 && bsr_span::lin_form(t, len, &c->cur_form[ck])) ? 1 : 0;
  if (c->cur_fmask.size() != c->cur_form.size()) c->cur_fmask.assign(c->cur_form.size(), ~0ull);
  c->cur_fmask[ck] = (t && len > 0) ? bsr_span::feature_mask(t, len) : ~0ull;
  auto b = std::make_shared<bsr_span::SpanBasis>();
  b->build(std::vector<bsr_span::LinForm>(c->cur_form.begin() + (size_t)chain * c->K, c->cur_form.begin() + (size_t)(chain + 1) * c->K),
           std::vector<char>(c->cur_form_ok.begin() + (size_t)chain * c->K, c->cur_form_ok.begin() + (size_t)(chain + 1) * c->K));
  b->feat_mask = 0;
  for (int j = 0; j < c->K; ++j) b->feat_mask |= c->cur_fmask[(size_t)chain * c->K + j];
  c->span[chain] = b;
}

int ensure_input(bsr_ctx* c, BatchSlot& s, size_t stream_words) {
  const size_t need = s.off_streams + stream_words * 8;
  if (need <= s.in_cap) return BSR_OK;
  const size_t cap = std::max(need, std::max(s.off_streams + (size_t)64 * 1024, s.in_cap * 2));
  if (s.d_in) HIPCHK(c, hipFree(s.d_in));
  if (s.h_in) HIPCHK(c, hipHostFree(s.h_in));
  s.d_in = nullptr;
  s.h_in = nullptr;
  HIPCHK(c, hipMalloc((void**)&s.d_in, cap));
  poison(s.d_in, cap);
  HIPCHK(c, hipHostMalloc((void**)&s.h_in, cap));
  s.in_cap = cap;
  return BSR_OK;
}

static LaunchGeom geometry(const bsr_ctx* c, const BatchSlot& s, int P) {
  LaunchGeom g;
  g.rb_rows = s.rb_rows;
  g.n_rb = (int)((c->N + g.rb_rows - 1) / g.rb_rows);
  const int want_pg = std::max(1, c->target_wgs / g.n_rb);       // proposal groups wanted
  int pg = (P + want_pg - 1) / want_pg;                            // proposals per workgroup
  if (P >= BSR_WG_WAVES) pg = (pg + BSR_WG_WAVES - 1) / BSR_WG_WAVES * BSR_WG_WAVES;
  pg = std::max(1, std::min(pg, P));
  g.pg = pg;
  g.n_pg = (P + pg - 1) / pg;
  // work-queue launch: one resident set of workgroups (5 per CU fit at <= 102 VGPRs), never more than there are tasks
  const int64_t tasks = (int64_t)P * g.n_rb;
  const int64_t want = std::min<int64_t>((int64_t)c->n_cu * c->wgs_per_cu, (tasks + BSR_WG_WAVES - 1) / BSR_WG_WAVES);
  g.dyn_wgs = (int)std::max<int64_t>(8, (want + 7) / 8 * 8);
  return g;
}

static int ensure_partials(bsr_ctx* c, BatchSlot& s, const LaunchGeom& g, int P, int spill_slots) {
  const size_t recs = (size_t)P * g.n_rb;
  if (recs > s.part_cap) {
    HIPCHK(c, hipStreamSynchronize(s.stream));
    if (s.part1) HIPCHK(c, hipFree(s.part1));
    if (s.part2) HIPCHK(c, hipFree(s.part2));
    s.part1 = s.part2 = nullptr;
    const size_t cap = recs + recs / 2;
    HIPCHK(c, hipMalloc((void**)&s.part1, cap * BSR_P1_WORDS * sizeof(double)));
    poison(s.part1, cap * BSR_P1_WORDS * sizeof(double));
    // K <= 3: the finalise step rides behind the residual pass in the same launch (its last workgroup to arrive runs
    // it).  The residual sums and the arrival counter then live in UNCACHED device memory: a store is visible once it
    // is acknowledged and a load never sees a cache, so the hand-over needs no cache write-back or invalidate under
    // the other batches' row passes (with agent-scope fences the fused launch measured 4 us per step slower than two
    // launches; this way it is 0.3 us faster).  8 B of counter behind the records.
    s.part2_uncached = false;
    if (residual_can_fuse_finalize(c->K) && c->no_lds &&
        hipExtMallocWithFlags((void**)&s.part2, cap * BSR_P2_WORDS * sizeof(double) + 64, hipDeviceMallocUncached) == hipSuccess) {
      s.part2_uncached = true;
    } else {
      (void)hipGetLastError();
      HIPCHK(c, hipMalloc((void**)&s.part2, cap * BSR_P2_WORDS * sizeof(double) + 64));
    }
    // on the slot's stream: the slot streams do not wait for the null stream (hipStreamNonBlocking), and a fill that
    // is still running when the batch's residual pass writes its records would wipe them
    HIPCHK(c, hipMemsetAsync(s.part2, 0, cap * BSR_P2_WORDS * sizeof(double) + 64, s.stream));
    s.part_cap = cap;
  }
  if (spill_slots > 0) {
    const size_t wgs = std::max((size_t)((g.n_rb + 7) / 8 * 8) * g.n_pg, (size_t)g.dyn_wgs);
    const size_t need = wgs * BSR_WG_WAVES * spill_slots * BSR_WAVE * 8 * c->esz;
    if (need > s.spill_cap) {
      HIPCHK(c, hipStreamSynchronize(s.stream));
      if (s.spill) HIPCHK(c, hipFree(s.spill));
      s.spill = nullptr;
      HIPCHK(c, hipMalloc(&s.spill, need));
      poison(s.spill, need);
      s.spill_cap = need;
    }
  }
  return BSR_OK;
}

static void fill_desc_tape(PropDesc* D, const TapeLoc& L) {
  memset(D, 0, sizeof *D);
  D->code_off = L.code_off;
  D->n_nodes = L.n_stream;
  D->feat_off = L.feat_off;
  D->ln_off = L.ln_off;
  D->spill_need = std::max(0, L.max_sp - 1 - 2);  // sized for the smallest register stack (2 slots at 8 rows/lane)
  D->max_sp = L.max_sp;
  D->cost = L.cost;
  D->chain = L.acc_only;
  D->grp = L.grp;
  D->n_term = L.nt;
  D->n_ln = L.nl;
}

static void fill_eval_desc(bsr_ctx* c, PropDesc* D, const TapeLoc& L, void* zout) {
  fill_desc_tape(D, L);
  D->mode = BSR_MODE_EVAL;
  D->nq = 0;
  D->K = c->K;
  D->qbase = nullptr;
  D->zout = zout;
  D->s = 1.0;
  D->sigma = 1.0;
}

template <typename T>
static void fill_row_args(bsr_ctx* c, BatchSlot& s, const LaunchGeom& g, RowPassArgs<T>* a, const PropDesc* desc,
                          int P, int spill_slots, int residual) {
  const uint64_t* codes = s.d_streams();
  const uint64_t* feats = codes + s.code_words;
  a->g = g;
  a->Xt = (const T*)c->Xt;
  a->y = c->has_y ? (const T*)c->y : nullptr;
  a->ld = c->ld;
  a->N = c->N;
  a->codes = codes;
  a->feats = feats;
  a->lnp = reinterpret_cast<const double*>(feats + s.feat_words);
  a->desc = desc;
  a->coef = s.d_coef;
  a->P = P;
  a->feat_list = s.use_lds ? s.d_feat() : nullptr;
  a->nF = s.nF;
  a->part = residual ? s.part2 : s.part1;
  a->spill = spill_slots ? s.spill : nullptr;
  a->spill_slots = spill_slots;
  a->rows_per_lane = c->rows_per_lane;
  if (!residual) {
    const uint32_t q = s.queue_seq++;
    a->queue = s.queue + (size_t)(q % BSR_QUEUE_SETS) * BSR_QUEUE_SET_INTS;
    a->queue_clear = s.queue + (size_t)((q + BSR_QUEUE_SETS / 2) % BSR_QUEUE_SETS) * BSR_QUEUE_SET_INTS;
  } else {
    a->queue = s.flag_cur();
    a->queue_clear = nullptr;
  }
}

static void launch_row_pass(bsr_ctx* c, BatchSlot& s, const LaunchGeom& g, const PropDesc* desc, int P,
                            int spill_slots, int nq, int residual, hipStream_t st = nullptr,
                            const FinArgs* fin = nullptr) {
  if (!st) st = s.stream;
  FinArgs none;
  memset(&none, 0, sizeof none);
  if (c->dtype == BSR_DTYPE_F64) {
    RowPassArgs<double> a;
    fill_row_args<double>(c, s, g, &a, desc, P, spill_slots, residual);
    a.fin = fin ? *fin : none;
    launch_rows<double>(st, a, nq, residual);
  } else {
    RowPassArgs<float> a;
    fill_row_args<float>(c, s, g, &a, desc, P, spill_slots, residual);
    a.fin = fin ? *fin : none;
    launch_rows<float>(st, a, nq, residual);
  }
}

// Everything a staged batch puts on the GPU: upload, row pass, the kernels behind it, the event the waiter blocks on.
struct TailJob {
  int slot, P, spill_slots, nq;
  LaunchGeom g;
  bool scoring, maybe_tile;
  bool restage;   // false: a rescoring run over descriptors of the batch already staged (groups and streams stand)
  bool deferred;  // the batch's staging itself (stage_submitted) still to be done, from the slot's copies of the inputs
  double rank_floor;
};
static void launcher_push(bsr_ctx* c, const TailJob& job);

// The two steps behind k_solve for the proposals it flagged: the residual pass (w = s z - Q c measured directly) and
// the finalise step -- fused into the residual pass's last workgroup where that kernel has the registers for the
// solver (K <= 3) and the hand-over memory is uncached (ensure_partials): one launch fewer.
static void launch_flagged_tail(bsr_ctx* c, BatchSlot& s, const LaunchGeom& g, int P, int spill_slots, int nq, double rank_floor,
                                int& rc) {
  hipStream_t st = s.stream;
  const bool fuse_fin = s.part2_uncached && residual_can_fuse_finalize(c->K) && c->no_lds;
  FinArgs fin;
  memset(&fin, 0, sizeof fin);
  if (fuse_fin) {
    fin.ck = c->d_ck; fin.out = s.h_out; fin.mh = s.d_mh; fin.rank_floor = rank_floor;
    fin.arrive = reinterpret_cast<int32_t*>(s.part2 + s.part_cap * BSR_P2_WORDS);   // counter behind the uncached records
  }
  launch_row_pass(c, s, g, s.d_desc(), P, spill_slots, nq, 1, st, fuse_fin ? &fin : nullptr);
  if (s.timed > 1) {
    const hipError_t e = hipEventRecord(s.ev[3], st);
    if (e != hipSuccess && rc == BSR_OK) { set_err(c, (std::string("hipEventRecord: ") + hipGetErrorString(e)).c_str()); rc = BSR_E_HIP; }
  }
  if (!fuse_fin)
    launch_finalize(st, s.d_desc(), c->d_ck, s.d_coef, P, g.n_rb, s.part2, c->N, s.h_out, rank_floor, s.flag_cur(), s.d_mh,
                    c->K <= 4 ? 1 : 16);
}

// Which candidates of the batch lie in the span of their chain's current columns by construction (bsr_span.h): the
// tree they would replace again, the same with a negation moved, a linear combination of current trees.  k_solve then
// takes w = 0 without the residual step -- if its own one-pass figure agrees.  fp32 columns: only repeats up to sign
// (bit-identical columns); a combination's rounding residue is above the cut there and goes through the residual step.
static void mark_in_span(bsr_ctx* c, BatchSlot& s, int P) {
  PropDesc* hd = s.h_desc();
  if ((int)s.off_copy.size() < P + 1) return;
  for (int i = 0; i < P; ++i) {
    PropDesc& D = hd[i];
    if (D.mode != BSR_MODE_SCORE || D.ck < 0 || (size_t)D.ck >= s.span_snap.size()) continue;
    const bsr_span::SpanBasis* b = s.span_snap[D.ck].get();
    if (!b) continue;
    // (a candidate that reads a feature none of the chain's current trees reads: not in their span but by an exact
    // cancellation -- two thirds of the real mix at d = 10, K = 3, and the canonical form is what this step costs)
    if (bsr_span::feature_mask(s.rows_copy.data() + s.off_copy[i], s.off_copy[i + 1] - s.off_copy[i]) & ~b->feat_mask) {
      D.self_dup = 0;
      continue;
    }
    bsr_span::LinForm f;
    if (!bsr_span::lin_form(s.rows_copy.data() + s.off_copy[i], s.off_copy[i + 1] - s.off_copy[i], &f)) continue;
    const bool rep = D.k >= 0 && (size_t)D.k < b->forms.size() && b->known[D.k] && bsr_span::same_up_to_sign(f, b->forms[D.k]);
    D.self_dup = (rep || (c->dtype == BSR_DTYPE_F64 && b->in_span(f))) ? 1 : 0;
  }
}

static int stage_submitted(bsr_ctx* c, BatchSlot& s, int B, TailJob* job);
static hipError_t use_device_fwd(const bsr_ctx* c);

static void publish_direct(bsr_ctx* c, BatchSlot& s, int rc, long long t_issue0);

static int issue_batch(bsr_ctx* c, BatchSlot& s, const TailJob& j_in) {
  const long long t_issue0 = host_now();
  TailJob j = j_in;
  if (j.deferred) {
    // the caller only validated and copied its inputs (bsr_internal_submit_mh): streams, descriptors and launch geometry
    // here, in front of the launches, on this thread
    const int rc0 = stage_submitted(c, s, j.P, &j);
    if (rc0 != BSR_OK) {   // nothing was launched: the waiter sees the error
      s.tail_rc = rc0;
      s.tail_gen.store(s.tail_wanted, std::memory_order_release);
      return rc0;
    }
  }
  hipStream_t st = s.stream;
  int rc = BSR_OK;
  auto step = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == BSR_OK) {
      set_err(c, (std::string(what) + ": " + hipGetErrorString(e)).c_str());
      rc = BSR_E_HIP;
    }
  };
  // the tile pass's half of the staging (groups, LDS slot maps, schedule, tape records), here with the launches: off
  // the caller's thread where the context has submission threads
  bool tile = false;
  TileGeom tg;
  memset(&tg, 0, sizeof tg);
  const long long tq0 = g_host_prof ? host_now() : 0;
  if (j.scoring && j.restage && c->selfdup) mark_in_span(c, s, j.P);
  if (j.scoring && c->solve_exact) {   // bit 1 of self_dup: no fast tier for this proposal (csrc/bsr_solve.h)
    PropDesc* hd = s.h_desc();
    for (int i = 0; i < j.P; ++i) hd[i].self_dup |= 2;
  }
  const long long tq1 = g_host_prof ? host_now() : 0;
  long long tq2 = tq1;
  if (j.maybe_tile) {
    if (j.restage) stage_tile(c, s, j.P);
    tq2 = g_host_prof ? host_now() : 0;
    if (s.tile) {
      rc = build_tile_launch(c, s, j.P, &tg);
      tile = rc == BSR_OK;
    }
  }
  if (g_host_prof) {
    g_ns_part[7].fetch_add(tq1 - tq0, std::memory_order_relaxed);
    g_ns_part[8].fetch_add(tq2 - tq1, std::memory_order_relaxed);
    g_ns_part[9].fetch_add(host_now() - tq2, std::memory_order_relaxed);
  }
  const int n_part = tile ? tg.n_part : j.g.n_rb;   // partial records per proposal that k_solve reduces
  if (j.maybe_tile && !tile) {   // the work-queue pass after all: its order of the tapes
    PropDesc* hd = s.h_desc();
    cost_order(s.order_tmp, s.order_keys, j.P, [&](int i) { return hd[i].cost; });
    for (int i = 0; i < j.P; ++i) hd[i].order = s.order_tmp[i];
  }
  long long t_part = g_host_prof ? host_now() : 0;
  auto part = [&](int i) {
    if (!g_host_prof) return;
    const long long t = host_now();
    g_ns_part[i].fetch_add(t - t_part, std::memory_order_relaxed);
    t_part = t;
  };
  if (g_host_prof) g_ns_part[0].fetch_add(t_part - t_issue0, std::memory_order_relaxed);
  const size_t in_bytes = tile ? s.off_recs + s.recs_bytes : s.off_streams + (s.code_words + s.feat_words + s.ln_words) * 8;
  if (rc != BSR_OK) {   // nothing was launched: the waiter sees the error
    s.tail_rc = rc;
    s.tail_gen.store(s.tail_wanted, std::memory_order_release);
    return rc;
  }
  // Direct dispatch (bsr_aql.h): the launch functions below append to the slot's packet list instead of calling the HIP
  // runtime, and the list goes to one of the context's own queues with a single doorbell write.  Only for a slot
  // whose stream has nothing in flight, with the input block written through the BAR; batches timed kernel by kernel
  // (bsr_set_profiling(2)) keep the stream: their events live there; a batch that times its row pass only reads the
  // packet processor's timestamps of that dispatch.
  bool use_aql = c->aql && !c->aql_off && j.scoring && s.timed <= 1 && c->bar_write && !s.stream_dirty && s.aql.signal != 0;
  hipStream_t s0 = s.stream;
  {
    // upload on the slot's stream
    if (c->bar_write && !s.stream_dirty) {
      // the input block goes into device memory by plain stores through the PCIe BAR (write-combined, ~1 us for
      // 16 KB) instead of a copy command: one HIP call fewer per batch.  The slot's previous batch has been waited
      // for, so nothing on the device reads the block now; the stores are globally ordered before the doorbell write
      // of the launch below (fence, then posted writes in order), and every kernel start invalidates the caches
      memcpy(s.d_in, s.h_in, in_bytes);
      if (!use_aql) __builtin_ia32_sfence();   // drain the write-combining buffers before anything that rings the doorbell
      // ... and read the last word back through the same mapping: a read cannot pass the posted writes in front of
      // it, so when it returns they have reached the device (what the runtime does for device-resident kernel
      // arguments; 2 us on the submission thread: -0.8 % at C2 with two threads, -4 % with one)
      if (in_bytes >= 8 && !use_aql) {   // (direct dispatch: the kernel arguments follow, one read-back behind both)
        const volatile uint64_t* last = reinterpret_cast<const volatile uint64_t*>((const char*)s.d_in + ((in_bytes - 8) & ~(size_t)7));
        s.bar_readback = *last;
      }
      std::atomic_thread_fence(std::memory_order_seq_cst);
    } else {
      // (also when bsr_commit or a rescore left work on the slot's stream that nobody waited for -- it reads or writes
      // the device block: the copy command is ordered behind it, host stores would not be)
      step(hipMemcpyAsync(s.d_in, s.h_in, in_bytes, hipMemcpyHostToDevice, s0), "hipMemcpyAsync");
    }
    s.stream_dirty = false;   // this batch's completion covers everything before it on the stream
    part(1);
  }
  // every launch of the batch, in stream order (or, collected for direct dispatch, in packet order)
  auto launches = [&]() {
    if (s.timed && !aql_target()) step(hipEventRecord(s.ev[0], s0), "hipEventRecord");
    if (tile) {
      const uint64_t* codes = s.d_streams();
      const uint64_t* feats = codes + s.code_words;
      const double* lnp = reinterpret_cast<const double*>(feats + s.feat_words);
      const uint64_t* feats_lds = reinterpret_cast<const uint64_t*>(lnp + s.ln_words);
      auto fill = [&](auto& a) {
        using TT = typename std::remove_reference<decltype(a)>::type;
        a.g = tg;
        a.colsrc = (decltype(TT::colsrc))s.d_cols();
        a.cols_stride = s.cols_stride;
        a.N = c->N; a.codes = codes; a.feats = feats_lds; a.lnp = lnp; a.desc = s.d_desc(); a.sched = s.d_sched();
        a.part = s.part1; a.P = j.P; a.K = c->K;
        a.stamps = c->d_stamps ? c->d_stamps + (size_t)(c->stamp_seq.fetch_add(1, std::memory_order_relaxed) % (uint32_t)c->stamp_ring) * c->stamp_block_words
                               : nullptr;
        a.srec = s.tile_stream ? reinterpret_cast<const StreamRec*>(s.d_in + s.off_recs + s.srec_off) : nullptr;
        a.tprog = (s.tprog_off != 0 && tg.per_group > 0) ? reinterpret_cast<const TileProg*>(s.d_in + s.off_recs + s.tprog_off) : nullptr;
        a.split_stage = c->tile_split;
        for (int i = 0; i < 8; ++i) a.grp_nF[i] = s.grp_nF[i];
        // the groups' column tables inside the argument block: 128 pointers dealt to 1, 2 or 4 groups
        constexpr int n_arg = BSR_TILE_ARG_GROUPS * BSR_TILE_ARG_COLS;
        a.arg_stride = tg.T == 1 ? n_arg : tg.T == 2 ? n_arg / 2 : BSR_TILE_ARG_COLS;
        a.cols_in_args = (tg.T <= BSR_TILE_ARG_GROUPS && tg.ncols <= a.arg_stride) ? 1 : 0;
        auto* flat = &a.cols[0][0];
        for (int i = 0; i < n_arg; ++i) flat[i] = nullptr;
        if (a.cols_in_args)
          for (int gi = 0; gi < tg.T; ++gi)
            for (int i = 0; i < tg.ncols; ++i)
              flat[gi * a.arg_stride + i] = (decltype(a.cols[0][0]))s.h_cols()[(size_t)gi * s.cols_stride + i];
      };
      if (c->dtype == BSR_DTYPE_F64) {
        TileArgs<double> a;
        fill(a);
        if (s.tile_stream) launch_stream(s0, a, s.tile_deep2);
        else if (a.tprog) launch_tile_asm(s0, a);
        else launch_tile<double>(s0, a);
      } else {
        TileArgs<float> a;
        fill(a);
        if (s.tile_stream) launch_stream_f32(s0, a);
        else launch_tile<float>(s0, a);
      }
    } else {
      launch_row_pass(c, s, j.g, s.d_desc(), j.P, j.spill_slots, j.nq, 0);
    }
    if (s.timed && !aql_target()) step(hipEventRecord(s.ev[1], s0), "hipEventRecord");
    part(2);
  // results go straight into the slot's pinned host block (device-visible): no download command behind the kernels
  launch_solve(st, s.d_desc(), c->d_ck, j.P, n_part, s.part1, c->N, s.d_coef, s.h_out, j.rank_floor, s.flag_cur(), s.d_mh,
               s.flag_other());
  part(3);
  if (s.timed > 1) step(hipEventRecord(s.ev[2], st), "hipEventRecord");
  if (j.scoring) launch_flagged_tail(c, s, j.g, j.P, j.spill_slots, j.nq, j.rank_floor, rc);
  else if (s.timed > 1) step(hipEventRecord(s.ev[3], st), "hipEventRecord");
  if (j.scoring && s.n_spans > 0)   // the scalar tail of newProp and the first-event scan, one event per chain span
    launch_events(st, s.d_mh, s.d_terms(), s.d_mhflags(), s.d_spans(), s.n_spans, c->K, s.h_ev);
  if (s.timed > 1) step(hipEventRecord(s.ev[4], st), "hipEventRecord");
  part(5);
  };
  s.aql_pending = false;
  if (use_aql) {
    s.aqb->n = 0;
    s.aqb->failed = false;
    aql_target() = s.aqb.get();
    aql_target_device() = c->aql;
    launches();
    aql_target() = nullptr;
    aql_target_device() = nullptr;
    if (!s.aqb->failed && s.aqb->n > 0) {
      aql_stage_args(c->aql, &s.aql, *s.aqb);
      aql_flush_writes(s.aql.d_kernarg);
      if (aql_submit(c->aql, &s.aql, *s.aqb, s.timed != 0) == 0) {
        s.aql_pending = true;
        s.aql_items = s.aqb->n;
      }
    }
    if (!s.aql_pending) {   // a kernel without a descriptor, a queue in error: the stream from now on
      c->aql_off = true;
      use_aql = false;
      aql_flush_writes(s.d_in);
    }
  }
  if (!s.aql_pending) launches();
  if (j.scoring) (s.aql_pending ? c->n_direct : c->n_streamed).fetch_add(1, std::memory_order_relaxed);
  if (s.aql_pending) {
    part(6);
    publish_direct(c, s, rc, t_issue0);
    return rc;
  }
  // completion: a word in pinned host memory that the queue itself writes behind the batch's last kernel (a stream
  // write-value command: no kernel, no fence of ours), which the waiter polls -- or the event it blocks on
  if (s.h_done_word && c->done_word) {
    s.done_wanted = ++s.done_gen;
    hipError_t ew = hipStreamWriteValue32(st, s.h_done_word, s.done_wanted, 0);
    if (ew != hipSuccess) {   // (not supported here after all: the event, from now on)
      (void)hipGetLastError();
      c->done_word = 0;
      s.done_wanted = 0;
      step(hipEventRecord(s.done, st), "hipEventRecord");
    }
  } else {
    s.done_wanted = 0;
    step(hipEventRecord(s.done, st), "hipEventRecord");
  }
  part(6);
  s.tail_rc = rc;
  if (g_host_prof) {
    g_ns_issue.fetch_add(host_now() - t_issue0, std::memory_order_relaxed);
    g_n_issue.fetch_add(1, std::memory_order_relaxed);
  }
  s.tail_gen.store(s.tail_wanted, std::memory_order_release);
  return rc;
}

static void publish_direct(bsr_ctx* c, BatchSlot& s, int rc, long long t_issue0) {
  s.done_wanted = 0;
  s.tail_rc = rc;
  if (g_host_prof) {
    g_ns_issue.fetch_add(host_now() - t_issue0, std::memory_order_relaxed);
    g_n_issue.fetch_add(1, std::memory_order_relaxed);
  }
  s.tail_gen.store(s.tail_wanted, std::memory_order_release);
}

// Second submission thread of a context.  It spins for a while after its last job (a sleeping thread would add its
// wake-up time to every batch of a busy pipeline) and sleeps on the condition variable when the context goes quiet.
struct Launcher {
  std::vector<std::thread> ths;   // one by default; BSR_SUBMIT_THREADS more share the queue (jobs of different slots)
  std::mutex mu;
  std::condition_variable cv;
  std::deque<TailJob> q;
  std::atomic<int> n_queued{0};
  int asleep = 0;
  bool stop = false;
  int spin_us = 100;
};

static void launcher_main(bsr_ctx* c) {
  bsr_internal_place_thread();
  (void)hipSetDevice(c->device);
  Launcher* L = c->launcher;
  for (;;) {
    TailJob job;
    bool have = false;
    const auto t_idle = std::chrono::steady_clock::now();
    while (!have) {
      if (L->n_queued.load(std::memory_order_acquire) > 0) {
        std::lock_guard<std::mutex> lk(L->mu);
        std::deque<TailJob>& from = L->q;
        if (!from.empty()) {
          job = from.front();
          from.pop_front();
          L->n_queued.fetch_sub(1, std::memory_order_relaxed);
          have = true;
        }
      } else if (std::chrono::steady_clock::now() - t_idle > std::chrono::microseconds(L->spin_us)) {
        std::unique_lock<std::mutex> lk(L->mu);
        if (L->stop) return;
        if (L->q.empty()) {
          ++L->asleep;
          L->cv.wait(lk, [&] { return L->stop || !L->q.empty(); });
          --L->asleep;
          if (L->stop && L->q.empty()) return;
        }
      } else {
        __builtin_ia32_pause();
      }
    }
    (void)issue_batch(c, c->slot[job.slot], job);
  }
}

static void launcher_push(bsr_ctx* c, const TailJob& job) {
  Launcher* L = c->launcher;
  bool wake;
  {
    std::lock_guard<std::mutex> lk(L->mu);
    L->q.push_back(job);
    L->n_queued.fetch_add(1, std::memory_order_release);
    wake = L->asleep > 0;
  }
  if (wake) L->cv.notify_one();
}

// CPUs this process may use: the affinity mask, cut down to the cgroup's CPU quota (the GPU boxes show 256 CPUs and
// grant 16), shared by the ranks of a multi-process run.
double bsr_internal_cpu_budget();
static double cpu_budget() { return bsr_internal_cpu_budget(); }
__attribute__((visibility("hidden"))) double bsr_internal_cpu_budget() {
  // what the ranks of the node share (the CPUs they may all run on, the cgroup's quota) is divided among them; a CPU
  // set the rank has to itself (BSR_PIN=1) is not
  const int local_world = std::max(1, env_int("LOCAL_WORLD_SIZE", env_int("WORLD_SIZE", 1)));
  double n = 1e9, own = 1e9;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) {
    if (g_pinned.load()) own = (double)CPU_COUNT(&set);
    else n = (double)CPU_COUNT(&set);
  }
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
    char q[32];
    double per = 0;
    if (fscanf(f, "%31s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) n = std::min(n, atof(q) / per);
    fclose(f);
  } else if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
    double quota = -1, per = 0;
    if (fscanf(fq, "%lf", &quota) != 1) quota = -1;
    fclose(fq);
    if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(fp, "%lf", &per) != 1) per = 0;
      fclose(fp);
    }
    if (quota > 0 && per > 0) n = std::min(n, quota / per);
  }
  return std::min(own, n / local_world);
}

static void launcher_start(bsr_ctx* c) {
  // Submission threads need cores of their own.  Two where the budget allows (HIP calls on different streams run in
  // parallel: 26-33 -> 21.6 us per step on a box whose CPU needs 25-33 us for a batch's six calls), one on a tight
  // budget -- down to two CPUs per rank, the share of eight ranks under a 16-CPU quota: the caller stages batch i + 1
  // while the thread issues batch i's calls, 15 instead of 27 us per step -- none below that (the caller then issues
  // its launches itself).  BSR_SUBMIT_THREAD=0 / 2: never / always; BSR_SUBMIT_THREADS=n: that many.
  const double cpus = cpu_budget();
  const int mode = env_int("BSR_SUBMIT_THREAD", 1);
  if (mode == 0 || (cpus < 2.0 && mode < 2)) return;
  c->launcher = new Launcher;
  c->launcher->spin_us = std::max(0, env_int("BSR_SUBMIT_SPIN_US", 100));
  const int n_th = std::max(1, std::min(4, env_int("BSR_SUBMIT_THREADS", cpus >= 6.0 ? 2 : 1)));
  for (int i = 0; i < n_th; ++i) c->launcher->ths.emplace_back(launcher_main, c);
}

static void launcher_stop(bsr_ctx* c) {
  Launcher* L = c->launcher;
  if (!L) return;
  {
    std::lock_guard<std::mutex> lk(L->mu);
    L->stop = true;
  }
  L->cv.notify_all();
  for (auto& th : L->ths)
    if (th.joinable()) th.join();
  delete L;
  c->launcher = nullptr;
}

// The launch geometry, partial-record buffers and tape order of the P descriptors staged in slot `s`: what issue_batch needs.
static int prepare_job(bsr_ctx* c, BatchSlot& s, int P, bool scoring, bool restage, TailJob* jobp) {
  PropDesc* hd = s.h_desc();
  int spill_slots = 0;
  for (int i = 0; i < P; ++i) spill_slots = std::max(spill_slots, hd[i].spill_need);
  LaunchGeom g = geometry(c, s, P);
  // whether the batch takes the tile pass is decided where its launches are issued (stage_tile): room for either
  const bool maybe_tile = scoring && s.tile_possible;
  {
    LaunchGeom gp = g;
    if (maybe_tile) gp.n_rb = std::max(g.n_rb, c->tile_slices + c->tile_left);   // partial records per proposal
    int rc0 = ensure_partials(c, s, gp, P, spill_slots);
    if (rc0 != BSR_OK) return rc0;
  }
  // work-queue order: heaviest tapes first (stable, so equal costs keep batch order); a batch that may take the tile
  // pass gets its order where that is decided (issue_batch)
  s.order_n = -1;
  if (!maybe_tile) {
    cost_order(s.order_tmp, s.order_keys, P, [&](int i) { return hd[i].cost; });
    for (int i = 0; i < P; ++i) hd[i].order = s.order_tmp[i];
  }
  const int nq = (hd[0].mode == BSR_MODE_SCORE) ? hd[0].nq : 0;
  s.timed = c->prof;  // the level in force when the batch was enqueued decides which events exist at wait time
  TailJob& job = *jobp;
  job.slot = (int)(&s - c->slot);
  job.P = P; job.g = g; job.spill_slots = spill_slots; job.nq = nq; job.scoring = scoring;
  job.maybe_tile = maybe_tile;
  job.restage = restage;
  job.deferred = false;
  // f32 columns: an exact duplicate leaves a residual of a few eps_f32; keep the gate's verdict on those
  job.rank_floor = (c->dtype == BSR_DTYPE_F32) ? 32.0 * 1.1920929e-7 : 0.0;
  s.flag_par ^= 1;   // this batch's list of flagged proposals; its k_solve empties the other one
  return BSR_OK;
}

// Enqueues upload + kernels + result download for the P descriptors staged in slot `s`.
static int enqueue(bsr_ctx* c, BatchSlot& s, int P, bool scoring, bool restage = true) {
  if (c->poisoned.load(std::memory_order_relaxed))
    return fail(c, BSR_E_STATE, "the context is poisoned: a directly dispatched batch never completed (destroy the context)");
  TailJob job;
  int rc = prepare_job(c, s, P, scoring, restage, &job);
  if (rc != BSR_OK) return rc;
  s.tail_rc = BSR_OK;
  s.tail_wanted = s.tail_gen.load(std::memory_order_relaxed) + 1;
  // Seven HIP calls per batch cost the host more than staging the batch does.  A scoring batch is handed to the
  // context's submission thread, which issues them while the caller returns to stage its next batch.
  if (c->launcher && scoring) {
    launcher_push(c, job);
  } else {
    rc = issue_batch(c, s, job);
    if (rc != BSR_OK) return rc;
  }
  s.P = P;
  s.pending = true;
  return BSR_OK;
}

// how long a directly dispatched batch may stay silent before its context is given up (BSR_DEBUG_AQL_TIMEOUT_MS: the
// poisoned-context path under test, tests/test_gpu_dispatch.py)
static const long long g_aql_timeout_ns = 1000000ll * (long long)env_int("BSR_DEBUG_AQL_TIMEOUT_MS", 60000);
static int wait_slot_impl(bsr_ctx* c, BatchSlot& s);
static int wait_slot(bsr_ctx* c, BatchSlot& s) {
  if (!g_host_prof || !s.pending) return wait_slot_impl(c, s);
  const long long t0 = host_now();
  const int rc = wait_slot_impl(c, s);
  g_ns_wait.fetch_add(host_now() - t0, std::memory_order_relaxed);
  g_n_wait.fetch_add(1, std::memory_order_relaxed);
  return rc;
}
static int wait_slot_impl(bsr_ctx* c, BatchSlot& s) {
  if (!s.pending) return BSR_OK;
  // the submission thread has issued this batch's launches (spin briefly, then give the core away: on a box with a
  // CPU quota two spinning threads throttle each other)
  for (int spins = 0; s.tail_gen.load(std::memory_order_acquire) != s.tail_wanted; ++spins) {
    if (spins < 4000) __builtin_ia32_pause();
    else std::this_thread::yield();
  }
  if (s.tail_rc != BSR_OK) {
    s.pending = false;
    return s.tail_rc;
  }
  bool was_direct = false;
  if (s.aql_pending) {
    was_direct = true;
    // the completion signal of the batch's last packet: a word in host memory the packet processor decrements
    const char* msg = nullptr;
    int st = 1;
    long long t_start = 0;
    for (long spins = 0; (st = aql_poll(c->aql, &s.aql, &msg)) > 0; ++spins) {
      if (g_aql_timeout_ns < 0) { st = -1; msg = "nothing (BSR_DEBUG_AQL_TIMEOUT_MS < 0: the first unfinished poll counts as silence)"; break; }
      if (spins < 20000) { __builtin_ia32_pause(); continue; }
      std::this_thread::yield();
      if ((spins & 1023) == 0) {   // a queue that stays silent for a minute: an error, not a hang of the caller
        const long long now = host_now();
        if (t_start == 0) t_start = now;
        else if (now - t_start > g_aql_timeout_ns) { st = -1; msg = "nothing for 60 s (the time limit of a batch)"; break; }
      }
    }
    s.aql_pending = false;
    if (st < 0) {
      // the packets may still be queued or running: nothing of this context may be staged over, resubmitted or freed
      // (the slot's buffers used to be reusable -- and freed by bsr_ctx_destroy -- under a batch still in flight)
      s.pending = false;
      c->aql_off = true;
      c->poisoned = true;
      return fail(c, BSR_E_HIP, (std::string("direct dispatch: the queue reported ") + (msg ? msg : "an error") +
                                 "; the context accepts no further batches (destroy it: its device buffers are left allocated)").c_str());
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else if (s.done_wanted != 0) {
    // the queue's own write behind the batch's last kernel: polled (a finished hipEventSynchronize costs the caller
    // 2.4 us per batch, a quarter of its time); a queue that has not written for two seconds is asked what happened
    const volatile uint32_t* w = s.h_done_word;
    long long t_start = 0;
    for (long spins = 0; *w != s.done_wanted; ++spins) {
      if (spins < 20000) { __builtin_ia32_pause(); continue; }
      std::this_thread::yield();
      if ((spins & 1023) == 0) {
        const long long now = host_now();
        if (t_start == 0) t_start = now;
        else if (now - t_start > 2000000000ll) {
          const hipError_t q = hipStreamQuery(s.stream);
          if (q != hipSuccess && q != hipErrorNotReady) { s.pending = false; HIPCHK(c, q); }
          t_start = now;
        }
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    HIPCHK(c, hipEventSynchronize(s.done));
    HIPCHK(c, hipGetLastError());
  }
  s.pending = false;
  if (s.timed && was_direct) {
    const double us = aql_row_us(c->aql, &s.aql, s.aql_items == 1);
    for (double& v : c->last_us) v = 0.0;
    c->last_us[0] = us > 0.0 ? us : 0.0;
  } else if (s.timed) {
    if (s.done_wanted != 0) HIPCHK(c, hipEventSynchronize(s.ev[s.timed > 1 ? 4 : 1]));   // (timed batches: the events' own clock)
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, s.ev[0], s.ev[1]));
    c->last_us[0] = ms * 1e3;
    if (s.timed > 1) {
      for (int i = 1; i < 4; ++i) {
        HIPCHK(c, hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]));
        c->last_us[i] = ms * 1e3;
      }
      HIPCHK(c, hipEventElapsedTime(&ms, s.ev[0], s.ev[4]));
      c->last_us[4] = ms * 1e3;
    }
  }
  return BSR_OK;
}

// The evaluate-only entry points are synchronous: they drain every slot and run in slot 0.
static int drain(bsr_ctx* c) {
  for (BatchSlot& s : c->slot) {
    int rc = wait_slot(c, s);
    if (rc != BSR_OK) return rc;
  }
  return BSR_OK;
}

static int ensure_zbuf(bsr_ctx* c) {
  if (c->zbuf) return BSR_OK;
  const size_t bytes = (size_t)c->ld * c->esz * c->max_batch;
  HIPCHK(c, hipMalloc(&c->zbuf, bytes));
  HIPCHK(c, hipMemsetAsync(c->zbuf, 0, bytes, c->stream));
  return BSR_OK;
}

extern "C" int bsr_eval_tapes(bsr_ctx* c, const bsr_node* rows, const int32_t* tape_off, int32_t n_tapes,
                              double* out_cols, double* maxabs, uint32_t* flags) {
  if (!c) return BSR_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = drain(c);
  if (rc != BSR_OK) return rc;
  BatchSlot& s = c->slot[0];
  std::vector<TapeLoc> loc;
  rc = stage_tapes(c, s, rows, tape_off, n_tapes, &loc);
  if (rc != BSR_OK) return rc;
  if (out_cols) {
    rc = ensure_zbuf(c);
    if (rc != BSR_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));  // zbuf's memset ran on the main stream
  }
  for (int i = 0; i < n_tapes; ++i)
    fill_eval_desc(c, &s.h_desc()[i], loc[i], out_cols ? col_ptr(c, c->zbuf, i) : nullptr);
  s.scored = false;
  rc = enqueue(c, s, n_tapes, false);
  if (rc == BSR_OK) rc = wait_slot(c, s);
  if (rc != BSR_OK) return rc;
  for (int i = 0; i < n_tapes; ++i) {
    if (maxabs) maxabs[i] = s.h_out[i].maxabs;
    if (flags) flags[i] = s.h_out[i].flags & (BSR_F_INF | BSR_F_NAN);
  }
  if (out_cols) {
    if (c->dtype == BSR_DTYPE_F64) {
      HIPCHK(c, hipMemcpy2DAsync(out_cols, (size_t)c->N * 8, c->zbuf, (size_t)c->ld * 8, (size_t)c->N * 8, n_tapes,
                                 hipMemcpyDeviceToHost, s.stream));
    } else {
      if (!c->d_stage) HIPCHK(c, hipMalloc((void**)&c->d_stage, (size_t)c->ld * 8));
      for (int i = 0; i < n_tapes; ++i) {
        launch_convert_out<float>(s.stream, (const float*)col_ptr(c, c->zbuf, i), c->d_stage, c->N);
        HIPCHK(c, hipMemcpyAsync(out_cols + (size_t)i * c->N, c->d_stage, (size_t)c->N * 8, hipMemcpyDeviceToHost,
                                 s.stream));
        HIPCHK(c, hipStreamSynchronize(s.stream));
      }
    }
    HIPCHK(c, hipStreamSynchronize(s.stream));
  }
  return BSR_OK;
}

static int chain_ok(bsr_ctx* c, int chain, int k) {
  if (chain < 0 || chain >= c->n_chains) return fail(c, BSR_E_ARG, "chain index out of range");
  if (k < 0 || k >= c->K) return fail(c, BSR_E_ARG, "tree index out of range");
  return BSR_OK;
}

// Fills the derived columns (kDerivedOps) behind X: d*kNumDerivedOps two-node tapes through the evaluation pass,
// each writing its own column of Xt.  Runs once at context creation, before anything else uses the slots.
static int build_derived(bsr_ctx* c) {
  BatchSlot& s = c->slot[0];
  const int total = c->d * kNumDerivedOps;
  std::vector<bsr_node> rows((size_t)2 * c->max_batch);
  std::vector<int32_t> off((size_t)c->max_batch + 1);
  std::vector<TapeLoc> loc;
  for (int t0 = 0; t0 < total; t0 += c->max_batch) {
    const int n = std::min(c->max_batch, total - t0);
    for (int i = 0; i < n; ++i) {
      const int m = (t0 + i) / c->d, f = (t0 + i) % c->d;
      bsr_node& a = rows[2 * i];
      bsr_node& b = rows[2 * i + 1];
      memset(&a, 0, sizeof a);
      memset(&b, 0, sizeof b);
      a.opcode = BSR_OP_TERMINAL; a.left = a.right = -1; a.feature = f;
      b.opcode = kDerivedOps[m]; b.left = 2 * i; b.right = -1; b.feature = -1;
      off[i] = 2 * i;
    }
    off[n] = 2 * n;
    // node links are tape-relative
    for (int i = 0; i < n; ++i) rows[2 * i + 1].left = 0;
    int rc = stage_tapes(c, s, rows.data(), off.data(), n, &loc);
    if (rc != BSR_OK) return rc;
    for (int i = 0; i < n; ++i) {
      const int m = (t0 + i) / c->d, f = (t0 + i) % c->d;
      fill_eval_desc(c, &s.h_desc()[i], loc[i], col_ptr(c, c->Xt, (int64_t)c->d * (1 + m) + f));
    }
    s.scored = false;
    rc = enqueue(c, s, n, false);
    if (rc == BSR_OK) rc = wait_slot(c, s);
    if (rc != BSR_OK) return rc;
  }
  c->derived_ready = true;
  return BSR_OK;
}

extern "C" int bsr_set_current(bsr_ctx* c, int32_t chain, int32_t k, const bsr_node* tape, int32_t len) {
  if (!c || !tape) return BSR_E_ARG;
  int rc = chain_ok(c, chain, k);
  if (rc != BSR_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  rc = drain(c);
  if (rc != BSR_OK) return rc;
  BatchSlot& s = c->slot[0];
  const int32_t off[2] = {0, len};
  std::vector<TapeLoc> loc;
  rc = stage_tapes(c, s, tape, off, 1, &loc);
  if (rc != BSR_OK) return rc;
  fill_eval_desc(c, &s.h_desc()[0], loc[0], col_ptr(c, c->cur, (int64_t)chain * c->K + k));
  s.scored = false;
  rc = enqueue(c, s, 1, false);
  if (rc == BSR_OK) rc = wait_slot(c, s);
  if (rc != BSR_OK) return rc;
  c->h_rin[chain].colmax[k] = s.h_out[0].maxabs;
  c->h_rin[chain].colflags[k] = s.h_out[0].flags & (BSR_F_INF | BSR_F_NAN);
  c->ready[chain] = 0;
  c->col_set[(size_t)chain * c->K + k] = 1;
  note_current_tree(c, chain, k, tape, len);
  return BSR_OK;
}

extern "C" int bsr_commit(bsr_ctx* c, int32_t chain, int32_t k, int32_t idx) {
  if (!c) return BSR_E_ARG;
  if (c->last_waited < 0) return fail(c, BSR_E_STATE, "bsr_commit: no scored batch to commit from");
  return bsr_internal_commit(c, c->last_waited, chain, k, idx);
}

void bsr_internal_feature_range(const bsr_ctx* c, const double** lo, const double** hi) {
  *lo = c->x_lo.data();
  *hi = c->x_hi.data();
}
void bsr_internal_lock(bsr_ctx* c) { c->mu.lock(); }
void bsr_internal_unlock(bsr_ctx* c) { c->mu.unlock(); }

// Commits candidate `idx` of the batch last scored on slot `si`.  Callers on several threads hold bsr_internal_lock
// across commit + refresh + fit (they share the main stream).
int bsr_internal_commit(bsr_ctx* c, int si, int32_t chain, int32_t k, int32_t idx) {
  int rc = chain_ok(c, chain, k);
  if (rc != BSR_OK) return rc;
  if (si < 0 || si >= BSR_SLOTS || !c->slot[si].scored)
    return fail(c, BSR_E_STATE, "bsr_commit: no scored batch to commit from");
  BatchSlot& s = c->slot[si];
  if (s.pending || s.waited_gen != s.gen)
    return fail(c, BSR_E_STATE, "bsr_commit: the slot of the last waited batch has been resubmitted since (commit before the next submit on it)");
  if (idx < 0 || idx >= s.P) return fail(c, BSR_E_STATE, "bsr_commit: index is not part of the last scored batch");
  HIPCHK(c, hipSetDevice(c->device));
  // Candidate columns are not kept by the scoring pass: re-run the still-staged tape straight into the chain cache.
  // The one-off descriptor goes to the spare entry behind the batch's own, so the batch stays intact.
  PropDesc D = s.h_desc()[idx];
  D.mode = BSR_MODE_EVAL;
  D.nq = 0;
  D.qbase = nullptr;
  D.zout = col_ptr(c, c->cur, (int64_t)chain * c->K + k);
  D.s = 1.0;
  D.order = 0;
  const int at = c->max_batch;
  s.h_desc()[at] = D;
  HIPCHK(c, hipMemcpyAsync(s.d_desc() + at, &s.h_desc()[at], sizeof(PropDesc), hipMemcpyHostToDevice, s.stream));
  const LaunchGeom g = geometry(c, s, 1);
  rc = ensure_partials(c, s, g, 1, D.spill_need);
  if (rc != BSR_OK) return rc;
  launch_row_pass(c, s, g, s.d_desc() + at, 1, D.spill_need, 0, 0);
  // the refresh that follows runs on the main stream: order it behind this column write
  HIPCHK(c, hipEventRecord(s.done, s.stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream, s.done, 0));
  s.stream_dirty = true;   // the re-run reads the slot's device input block and nobody waits for it here
  c->h_rin[chain].colmax[k] = s.h_out[idx].maxabs;
  c->h_rin[chain].colflags[k] = s.h_out[idx].flags & (BSR_F_INF | BSR_F_NAN);
  c->ready[chain] = 0;
  c->col_set[(size_t)chain * c->K + k] = 1;
  if ((size_t)idx + 1 < s.off_copy.size())
    note_current_tree(c, chain, k, s.rows_copy.data() + s.off_copy[idx], s.off_copy[idx + 1] - s.off_copy[idx]);
  else
    note_current_tree(c, chain, k, nullptr, 0);   // unknown: no shortcut for proposals on this chain's tree k
  return BSR_OK;
}

static int refresh_slow(bsr_ctx* c, int chain) {  // single-workgroup Gram-Schmidt path (robust for dependent siblings)
  const int K = c->K;
  void* cols = col_ptr(c, c->cur, (int64_t)chain * K);
  ChainB* dck = c->d_ck + (size_t)chain;
  const RefreshIn* rin = c->d_rin + chain;
  if (c->dtype == BSR_DTYPE_F64)
    launch_refresh_basis<double>(c->stream, (const double*)cols, (double*)col_ptr(c, c->Q, (int64_t)chain * K),
                                 (const double*)c->y, c->ld, c->N, K, rin->colmax, rin->colflags, dck);
  else
    launch_refresh_basis<float>(c->stream, (const float*)cols, (float*)col_ptr(c, c->Q, (int64_t)chain * K),
                                (const float*)c->y, c->ld, c->N, K, rin->colmax, rin->colflags, dck);
  return BSR_OK;
}

extern "C" int bsr_refresh(bsr_ctx* c, int32_t chain, bsr_chain_info* info) {
  if (!c) return BSR_E_ARG;
  int rc = chain_ok(c, chain, 0);
  if (rc != BSR_OK) return rc;
  if (!c->has_y) return fail(c, BSR_E_STATE, "bsr_refresh: context has no y");
  for (int k = 0; k < c->K; ++k)
    if (!c->col_set[(size_t)chain * c->K + k]) return fail(c, BSR_E_STATE, "bsr_refresh: a current column was never set");
  HIPCHK(c, hipSetDevice(c->device));
  const int K = c->K;
  ChainFitOut* dfit = c->d_fit + chain;
  ChainFitOut* dfit_i = c->d_fit_icpt + chain;
  void* cols = col_ptr(c, c->cur, (int64_t)chain * K);
  ChainB* dck = c->d_ck + (size_t)chain;
  HIPCHK(c, hipMemcpyAsync(c->d_rin + chain, &c->h_rin[chain], sizeof(RefreshIn), hipMemcpyHostToDevice, c->stream));
  if (c->fast_refresh) {
    void* Q = col_ptr(c, c->Q, (int64_t)chain * K);
    if (c->dtype == BSR_DTYPE_F64)
      launch_refresh_fast<double>(c->stream, (const double*)cols, (double*)Q, (const double*)c->y, c->ld, c->N, K,
                                  c->d_rin + chain, c->d_plan, c->d_rpart, dck, dfit, dfit_i);
    else
      launch_refresh_fast<float>(c->stream, (const float*)cols, (float*)Q, (const float*)c->y, c->ld, c->N, K,
                                 c->d_rin + chain, c->d_plan, c->d_rpart, dck, dfit, dfit_i);
    HIPCHK(c, hipMemcpyAsync(c->h_plan, c->d_plan, sizeof(RefreshPlan), hipMemcpyDeviceToHost, c->stream));
  } else {
    if (c->dtype == BSR_DTYPE_F64) {
      launch_chain_fit<double>(c->stream, (const double*)cols, (const double*)c->y, c->ld, c->N, K, 0, dfit);
      launch_chain_fit<double>(c->stream, (const double*)cols, (const double*)c->y, c->ld, c->N, K, 1, dfit_i);
    } else {
      launch_chain_fit<float>(c->stream, (const float*)cols, (const float*)c->y, c->ld, c->N, K, 0, dfit);
      launch_chain_fit<float>(c->stream, (const float*)cols, (const float*)c->y, c->ld, c->N, K, 1, dfit_i);
    }
    refresh_slow(c, chain);
  }
  HIPCHK(c, hipMemcpyAsync(&c->h_fit[chain], dfit, sizeof(ChainFitOut), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&c->h_fit_icpt[chain], dfit_i, sizeof(ChainFitOut), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&c->h_ck[(size_t)chain], dck, sizeof(ChainB), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  if (c->fast_refresh) {
    if (c->h_plan->fallback != 0) {  // (nearly) dependent columns: Cholesky-QR is not accurate enough, rebuild with Gram-Schmidt
      refresh_slow(c, chain);
      HIPCHK(c, hipMemcpyAsync(&c->h_ck[(size_t)chain], dck, sizeof(ChainB), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      HIPCHK(c, hipGetLastError());
    }
  }
  c->ready[chain] = 1;
  if (info) {
    const ChainFitOut& f = c->h_fit[chain];
    memset(info, 0, sizeof *info);
    info->sse_old = f.sse;
    info->scale_old = f.scale;
    for (int k = 0; k < K; ++k) {
      info->maxabs[k] = c->h_rin[chain].colmax[k];
      info->beta_old[k] = f.beta[k];
      info->colflags[k] = c->h_rin[chain].colflags[k];
    }
    info->rank_old = -2;
  }
  return BSR_OK;
}

// Stages and enqueues one batch on slot `si` (bsr_score_submit picks the slots round-robin; the native sampler's
// worker threads each own one).
int bsr_internal_submit(bsr_ctx* c, int si, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                        const int32_t* which_k, const double* sigma, int32_t B) {
  return bsr_internal_submit_mh(c, si, rows, tape_off, chain, which_k, sigma, B, nullptr, nullptr, nullptr, 0, false);
}

// The staging of a submitted batch from the slot's copies of its inputs (rows_copy / off_copy / sub_*): streams and tape
// groups (stage_tapes), descriptors, launch geometry.  On the caller's thread, or -- deferred -- on a submission thread in
// front of the batch's launches (issue_batch).
static int stage_submitted(bsr_ctx* c, BatchSlot& s, int B, TailJob* job) {
  const int K = c->K;
  const int32_t* chain = s.sub_chain.data();
  const int32_t* which_k = s.sub_k.data();
  std::vector<TapeLoc>& loc = s.loc_tmp;   // (kept with the slot: a fresh vector is an allocation per batch)
  const long long th0 = host_now();
  for (int32_t ch : s.batch_chains) s.chain_slot[ch] = -1;
  s.batch_chains.clear();
  for (int i = 0; i < B; ++i)
    if (s.chain_slot[chain[i]] < 0) {
      s.chain_slot[chain[i]] = (int32_t)s.batch_chains.size();
      s.batch_chains.push_back(chain[i]);
    }
  int rc = stage_tapes(c, s, s.rows_copy.data(), s.off_copy.data(), B, &loc, (int)s.batch_chains.size());
  if (rc != BSR_OK) return rc;
  const long long th1 = host_now();
  PropDesc* hd = s.h_desc();
  for (int i = 0; i < B; ++i) {
    PropDesc* D = &hd[i];
    fill_desc_tape(D, loc[i]);
    D->mode = BSR_MODE_SCORE;
    D->nq = K;
    D->k = which_k[i];
    D->K = K;
    D->ck = chain[i];
    D->qbase = col_ptr(c, c->Q, (int64_t)chain[i] * K);
    D->zout = nullptr;
    D->qslot = 0;   // (tile pass: set with the tape's group by stage_tile)
    D->s = s.sub_s[i];   // the chain's prescale for tree k as it stood when the batch was submitted
    D->sigma = s.sub_sigma[i];
  }
  const long long th2 = host_now();
  rc = prepare_job(c, s, B, true, true, job);
  if (g_host_prof) {
    g_ns_stage.fetch_add(th1 - th0, std::memory_order_relaxed);
    g_ns_desc.fetch_add(th2 - th1, std::memory_order_relaxed);
    g_ns_enq.fetch_add(host_now() - th2, std::memory_order_relaxed);
    g_n_sub.fetch_add(1, std::memory_order_relaxed);
  }
  return rc;
}

// `defer`: leave the staging to the submission thread that issues the batch's launches (a plain caller's thread is the
// pipeline's bottleneck: 4.4 of its 9.5 us per batch were staging); the caller still validates every tape, so
// an argument error comes back from the submit call itself.  What the deferred staging can still fail on (buffer
// growth, a schedule beyond its capacity) surfaces at the wait, like a failed launch.
int bsr_internal_submit_mh(bsr_ctx* c, int si, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                           const int32_t* which_k, const double* sigma, int32_t B, const double* terms8,
                           const int32_t* mhflags, const int32_t* span_off, int32_t n_spans, bool defer) {
  if (!c->has_y || c->K <= 0) return fail(c, BSR_E_STATE, "bsr_score_submit: context has no y / no chains");
  if (c->poisoned.load(std::memory_order_relaxed))
    return fail(c, BSR_E_STATE, "the context is poisoned: a directly dispatched batch never completed (destroy the context)");
  HIPCHK(c, use_device_fwd(c));
  for (int i = 0; i < B; ++i) {
    int rc = chain_ok(c, chain[i], which_k[i]);
    if (rc != BSR_OK) return rc;
    if (!c->ready[chain[i]]) return fail(c, BSR_E_STATE, "bsr_score_submit: chain not refreshed");
  }
  BatchSlot& s = c->slot[si];
  if (s.pending) return fail(c, BSR_E_STATE, "bsr_score_submit: every batch slot is in flight (wait first)");
  if (!rows || !tape_off || B <= 0) return fail(c, BSR_E_ARG, "null tapes / empty batch");
  if (B > c->max_batch) return fail(c, BSR_E_TOOBIG, "batch larger than max_batch");
  if (tape_off[0] != 0) return fail(c, BSR_E_ARG, "tape_off[0] must be 0");
  if (n_spans > 0) {
    if (!terms8 || !mhflags || !span_off || n_spans > B || span_off[0] != 0 || span_off[n_spans] != B)
      return fail(c, BSR_E_ARG, "bsr_score_submit_mh: bad terms / flags / spans");
    for (int j = 0; j < n_spans; ++j) {
      if (span_off[j + 1] <= span_off[j]) return fail(c, BSR_E_ARG, "bsr_score_submit_mh: empty or unordered span");
      for (int i = span_off[j]; i < span_off[j + 1]; ++i)
        if (chain[i] != chain[span_off[j]]) return fail(c, BSR_E_ARG, "bsr_score_submit_mh: a span mixes chains");
    }
  }
  defer = defer && c->launcher != nullptr && n_spans <= 0;
  if (defer) {   // (the staging validates as it goes; deferred, the caller still must)
    for (int i = 0; i < B; ++i) {
      int mx = 0;
      int rc = check_tape(c, rows + tape_off[i], tape_off[i + 1] - tape_off[i], &mx);
      if (rc != BSR_OK) return rc;
    }
  }
  // keep the batch's inputs: the staging reads them (perhaps on another thread, when the caller's arrays may be gone),
  // and bsr_commit makes one of the tapes a current tree (its canonical form is needed then)
  s.rows_copy.assign(rows, rows + tape_off[B]);
  s.off_copy.assign(tape_off, tape_off + B + 1);
  s.sub_chain.assign(chain, chain + B);
  s.sub_k.assign(which_k, which_k + B);
  s.sub_sigma.assign(sigma, sigma + B);
  s.sub_s.resize((size_t)B);
  for (int i = 0; i < B; ++i) {
    s.sub_s[i] = c->h_ck[chain[i]].s_k[which_k[i]];
    // (whether the candidate lies in the span of its chain's current columns by construction -- D->self_dup -- is worked
    // out where the batch's launches are issued, against the bases snapshotted here: mark_in_span)
    if (c->selfdup) {
      if (s.span_snap.size() != c->span.size()) s.span_snap.resize(c->span.size());
      if (s.span_snap[chain[i]].get() != c->span[chain[i]].get()) s.span_snap[chain[i]] = c->span[chain[i]];
    }
  }
  s.n_spans = 0;
  if (defer) {
    s.scored = true;
    ++s.gen;
    TailJob job;
    memset(&job, 0, sizeof job);
    job.slot = si;
    job.P = B;
    job.scoring = true;
    job.restage = true;
    job.deferred = true;
    s.tail_rc = BSR_OK;
    s.tail_wanted = s.tail_gen.load(std::memory_order_relaxed) + 1;
    launcher_push(c, job);
    s.P = B;
    s.pending = true;
    return BSR_OK;
  }
  TailJob job;
  int rc = stage_submitted(c, s, B, &job);
  if (rc != BSR_OK) return rc;
  if (n_spans > 0) {
    memcpy(s.h_terms(), terms8, sizeof(double) * 8 * (size_t)B);
    memcpy(s.h_mhflags(), mhflags, sizeof(int32_t) * (size_t)B);
    memcpy(s.h_spans(), span_off, sizeof(int32_t) * ((size_t)n_spans + 1));
    s.n_spans = n_spans;
  }
  s.scored = true;
  ++s.gen;
  s.tail_rc = BSR_OK;
  s.tail_wanted = s.tail_gen.load(std::memory_order_relaxed) + 1;
  if (c->launcher) {
    launcher_push(c, job);
  } else {
    rc = issue_batch(c, s, job);
    if (rc != BSR_OK) return rc;
  }
  s.P = B;
  s.pending = true;
  return BSR_OK;
}

extern "C" int bsr_score_submit(bsr_ctx* c, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                                const int32_t* which_k, const double* sigma, int32_t B, int32_t* ticket) {
  if (!c || !chain || !which_k || !sigma || !ticket) return BSR_E_ARG;
  const int si = c->next_slot;
  static const bool defer = env_int("BSR_DEFER_STAGE", 0) != 0;
  int rc = bsr_internal_submit_mh(c, si, rows, tape_off, chain, which_k, sigma, B, nullptr, nullptr, nullptr, 0, defer);
  if (rc != BSR_OK) return rc;
  *ticket = si;
  c->next_slot = (si + 1) % BSR_MAX_INFLIGHT;
  return BSR_OK;
}

// Waits for slot `ticket` only and copies its scores out; touches no other slot and no context-wide state (K > 1).
// the context's device for the calling thread: asked for only when another one is current (hipGetDevice reads a
// thread-local; hipSetDevice takes the runtime's locks -- twice per batch on a caller whose thread is the pipeline's bound)
static inline hipError_t use_device(const bsr_ctx* c) {
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur == c->device) return hipSuccess;
  return hipSetDevice(c->device);
}

int bsr_internal_wait(bsr_ctx* c, int ticket, bsr_score* out) {
  HIPCHK(c, use_device(c));
  BatchSlot& s = c->slot[ticket];
  if (!s.scored) return fail(c, BSR_E_STATE, "bsr_score_wait: nothing submitted under this ticket");
  int rc = wait_slot(c, s);
  if (rc != BSR_OK) return rc;
  s.waited_gen = s.gen;
  memcpy(out, s.h_out, sizeof(bsr_score) * s.P);
  return BSR_OK;
}

extern "C" int bsr_score_submit_mh(bsr_ctx* c, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                                   const int32_t* which_k, const double* sigma, int32_t B, const double* terms8,
                                   const int32_t* flags, const int32_t* span_off, int32_t n_spans, int32_t* ticket) {
  if (!c || !chain || !which_k || !sigma || !ticket || n_spans <= 0) return BSR_E_ARG;
  if (c->K == 1) return fail(c, BSR_E_STATE, "bsr_score_submit_mh: treeNum = 1 rescoring needs the plain submit/wait pair");
  const int si = c->next_slot;
  int rc = bsr_internal_submit_mh(c, si, rows, tape_off, chain, which_k, sigma, B, terms8, flags, span_off, n_spans, false);
  if (rc != BSR_OK) return rc;
  *ticket = si;
  c->next_slot = (si + 1) % BSR_MAX_INFLIGHT;
  return BSR_OK;
}

static hipError_t use_device_fwd(const bsr_ctx* c) { return use_device(c); }

int bsr_internal_wait_mh(bsr_ctx* c, int ticket, bsr_score* out, bsr_event* events) {
  HIPCHK(c, use_device(c));
  BatchSlot& s = c->slot[ticket];
  if (!s.scored || s.n_spans <= 0) return fail(c, BSR_E_STATE, "bsr_score_wait_mh: no MH batch under this ticket");
  int rc = wait_slot(c, s);
  if (rc != BSR_OK) return rc;
  s.waited_gen = s.gen;
  if (out) memcpy(out, s.h_out, sizeof(bsr_score) * s.P);
  memcpy(events, s.h_ev, sizeof(bsr_event) * s.n_spans);
  return BSR_OK;
}

extern "C" int bsr_score_wait_mh(bsr_ctx* c, int32_t ticket, bsr_score* out, bsr_event* events) {
  if (!c || !events || ticket < 0 || ticket >= BSR_MAX_INFLIGHT) return BSR_E_ARG;
  int rc = bsr_internal_wait_mh(c, ticket, out, events);
  if (rc == BSR_OK) c->last_waited = ticket;
  return rc;
}

extern "C" int bsr_score_wait(bsr_ctx* c, int32_t ticket, bsr_score* out) {
  if (!c || !out || ticket < 0 || ticket >= BSR_MAX_INFLIGHT) return BSR_E_ARG;
  int rc = bsr_internal_wait(c, ticket, out);
  if (rc != BSR_OK) return rc;
  BatchSlot& s = c->slot[ticket];
  const int B = s.P;
  c->last_waited = ticket;
  if (c->K == 1) {
    // no sibling column fixes the accumulation scale: rescore candidates whose |z|^2 left the double range
    std::vector<int> redo;
    for (int i = 0; i < B; ++i)
      if (out[i].flags & BSR_F_SCALE_RETRY) redo.push_back(i);
    if (!redo.empty()) {
      rc = drain(c);
      if (rc != BSR_OK) return rc;
      std::vector<PropDesc> keep(s.h_desc(), s.h_desc() + B);
      std::vector<bsr_score> batch_scores(s.h_out, s.h_out + B);  // the rescoring run reuses entries 0..redo-1
      double timing[5];
      memcpy(timing, c->last_us, sizeof timing);
      for (size_t j = 0; j < redo.size(); ++j) {
        PropDesc D = keep[redo[j]];
        int e = 0;
        std::frexp(out[redo[j]].maxabs, &e);
        e = std::max(-1000, std::min(1000, e));
        D.s = std::ldexp(1.0, -e);
        s.h_desc()[j] = D;
      }
      rc = enqueue(c, s, (int)redo.size(), true, false);   // (a descriptor keeps its tape group: the group's LDS slot map numbers its columns)
      if (rc == BSR_OK) rc = wait_slot(c, s);
      if (rc != BSR_OK) return rc;
      for (size_t j = 0; j < redo.size(); ++j) {
        out[redo[j]] = s.h_out[j];
        out[redo[j]].flags &= ~BSR_F_SCALE_RETRY;
        batch_scores[redo[j]] = out[redo[j]];
      }
      // the slot's result block goes back to batch order (bsr_commit reads max|z| and the flags of the accepted
      // candidate from it), and so do the batch's own descriptors, host and device
      memcpy(s.h_out, batch_scores.data(), sizeof(bsr_score) * B);
      memcpy(s.h_desc(), keep.data(), sizeof(PropDesc) * B);
      HIPCHK(c, hipMemcpyAsync(s.d_desc(), s.h_desc(), sizeof(PropDesc) * B, hipMemcpyHostToDevice, s.stream));
      s.stream_dirty = true;   // a copy into the device input block is still on its way
      s.P = B;
      memcpy(c->last_us, timing, sizeof timing);
    }
  }
  return BSR_OK;
}

extern "C" int bsr_score_batch(bsr_ctx* c, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                               const int32_t* which_k, const double* sigma, int32_t B, bsr_score* out) {
  if (!c || !out) return BSR_E_ARG;
  int32_t ticket = -1;
  int rc = bsr_score_submit(c, rows, tape_off, chain, which_k, sigma, B, &ticket);
  if (rc != BSR_OK) return rc;
  return bsr_score_wait(c, ticket, out);
}

extern "C" int bsr_fit_beta(bsr_ctx* c, int32_t chain, double* beta_out, double* rmse_out) {
  if (!c || !beta_out || !rmse_out) return BSR_E_ARG;
  int rc = chain_ok(c, chain, 0);
  if (rc != BSR_OK) return rc;
  if (!c->has_y) return fail(c, BSR_E_STATE, "bsr_fit_beta: context has no y");
  if (!c->ready[chain]) {  // the intercept fit is produced by the refresh pipeline
    rc = bsr_refresh(c, chain, nullptr);
    if (rc != BSR_OK) return rc;
  }
  const ChainFitOut& h = c->h_fit_icpt[chain];
  for (int j = 0; j <= c->K; ++j) beta_out[j] = h.beta_unscaled[j];
  *rmse_out = std::sqrt(h.sse / (double)c->N);
  return BSR_OK;
}

extern "C" int bsr_get_current(bsr_ctx* c, int32_t chain, double* out_cols) {
  if (!c || !out_cols) return BSR_E_ARG;
  int rc = chain_ok(c, chain, 0);
  if (rc != BSR_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->dtype == BSR_DTYPE_F64) {
    HIPCHK(c, hipMemcpy2DAsync(out_cols, (size_t)c->N * 8, col_ptr(c, c->cur, (int64_t)chain * c->K),
                               (size_t)c->ld * 8, (size_t)c->N * 8, c->K, hipMemcpyDeviceToHost, c->stream));
  } else {
    if (!c->d_stage) HIPCHK(c, hipMalloc((void**)&c->d_stage, (size_t)c->ld * 8));
    for (int k = 0; k < c->K; ++k) {
      launch_convert_out<float>(c->stream, (const float*)col_ptr(c, c->cur, (int64_t)chain * c->K + k), c->d_stage,
                                c->N);
      HIPCHK(c, hipMemcpyAsync(out_cols + (size_t)k * c->N, c->d_stage, (size_t)c->N * 8, hipMemcpyDeviceToHost,
                               c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BSR_OK;
}

extern "C" int bsr_yloglike_host(int device, int64_t N, int32_t K, const double* outputs, const double* y, double sigma,
                                 int32_t skipna, double* loglik, double* sse, double* scale, double* beta,
                                 int32_t* rank) {
  if (!outputs || !y || !loglik || K <= 0 || K > BSR_MAX_K) return BSR_E_ARG;
  // The K output columns play the role of X; tree k is the terminal "feature k".  Scoring the proposal that
  // replaces the last tree by itself yields ylogLike(y, outputs, sigma) through the regular kernels.
  bsr_ctx* c = nullptr;
  int rc = bsr_ctx_create(&c, device, N, K, outputs, y, K, 1, 1, BSR_DTYPE_F64);
  if (rc != BSR_OK) return rc;
  c->solve_exact = 1;   // ylogLike has no rank gate in front of it: a value for deficient inputs too (codes/funcs.py:1147-1174)
  bsr_node t;
  memset(&t, 0, sizeof t);
  t.opcode = BSR_OP_TERMINAL;
  t.left = t.right = -1;
  for (int k = 0; k < K && rc == BSR_OK; ++k) {
    t.feature = k;
    rc = bsr_set_current(c, 0, k, &t, 1);
  }
  bsr_chain_info info;
  if (rc == BSR_OK) rc = bsr_refresh(c, 0, &info);
  bsr_score sc;
  memset(&sc, 0, sizeof sc);
  if (rc == BSR_OK) {
    const int32_t off[2] = {0, 1};
    const int32_t ch = 0, kk = K - 1;
    t.feature = K - 1;
    rc = bsr_score_batch(c, &t, off, &ch, &kk, &sigma, 1, &sc);
  }
  if (rc != BSR_OK) {
    g_create_error = c->err;
    bsr_ctx_destroy(c);
    return rc;
  }
  double e = sc.sse, ll = sc.loglik;
  if (sc.flags & (BSR_F_INF | BSR_F_NAN)) {
    // every fitted value is NaN: Series.sum(skipna=True) -> 0.0, ndarray sum -> NaN (codes/funcs.py:1162)
    e = skipna ? 0.0 : NAN;
    ll = -e / (2 * sigma * sigma) - 0.5 * (double)N * std::log(2 * M_PI * sigma * sigma);
  }
  *loglik = ll;
  if (sse) *sse = e;
  if (scale) *scale = (sc.flags & BSR_F_NAN) ? NAN : sc.scale;
  if (beta) for (int k = 0; k < K; ++k) beta[k] = sc.beta[k];
  if (rank) *rank = sc.rank;
  bsr_ctx_destroy(c);
  return BSR_OK;
}

// Diagnostics (not part of the product surface): clock samples of the last tile launch, see bsr_tile.hip.
// out: [workgroups][16 waves][8] uint64; returns the number of workgroups, 0 if stamps are off (BSR_TILE_STAMPS unset).
extern "C" int bsr_debug_tile_stamps(bsr_ctx* c, unsigned long long* out, int32_t max_wgs, int32_t* geom5) {
  if (!c || !out) return BSR_E_ARG;
  if (!c->d_stamps) return 0;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  const int n = std::min<int>(max_wgs, c->tile_cus);
  if (hipMemcpy(out, c->d_stamps, (size_t)n * BSR_TILE_WAVES * BSR_TILE_STAMP_WORDS * sizeof(unsigned long long),
                hipMemcpyDeviceToHost) != hipSuccess) return BSR_E_HIP;
  if (geom5) {
    geom5[0] = c->tile_T; geom5[1] = c->tile_slices; geom5[2] = c->tile_bps; geom5[3] = c->tile_blocks;
    geom5[4] = c->tile_cus * 100 + 1;
  }
  return n;
}

// The whole ring (BSR_TILE_STAMPS=R): out[R][block_workgroups][16][8] uint64; info4 = {R, workgroup slots per block,
// workgroups of a launch, tile launches so far}.  Returns the number of blocks copied (<= max_blocks).
extern "C" int bsr_debug_tile_stamp_ring(bsr_ctx* c, unsigned long long* out, int32_t max_blocks, int32_t* info4) {
  if (!c || !out || !info4) return BSR_E_ARG;
  if (!c->d_stamps) return 0;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  const int n = std::min<int>(max_blocks, c->stamp_ring);
  if (hipMemcpy(out, c->d_stamps, (size_t)n * c->stamp_block_words * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
    return BSR_E_HIP;
  info4[0] = c->stamp_ring;
  info4[1] = (int32_t)(c->stamp_block_words / (BSR_TILE_WAVES * BSR_TILE_STAMP_WORDS));
  info4[2] = c->tile_cus;
  info4[3] = (int32_t)c->stamp_seq.load();
  return n;
}

extern "C" int bsr_batch_stats(const bsr_ctx* c, int32_t ticket, int32_t* stats4) {
  if (!c || !stats4 || ticket < 0 || ticket >= BSR_SLOTS) return BSR_E_ARG;
  const BatchSlot& s = c->slot[ticket];
  stats4[0] = s.stat_tapes; stats4[1] = s.stat_fast; stats4[2] = s.stat_chain; stats4[3] = s.stat_entries;
  return BSR_OK;
}

extern "C" int bsr_dispatch_info(const bsr_ctx* c, int64_t* info8) {
  if (!c || !info8) return BSR_E_ARG;
  for (int i = 0; i < 8; ++i) info8[i] = 0;
  info8[0] = (c->aql && !c->aql_off) ? 1 : 0;
  info8[1] = c->aql ? aql_n_queues(c->aql) : 0;
  info8[2] = c->n_direct.load();
  info8[3] = c->n_streamed.load();
  return BSR_OK;
}

extern "C" int bsr_ctx_info(const bsr_ctx* c, int32_t* info8) {
  if (!c || !info8) return BSR_E_ARG;
  info8[0] = c->launcher ? (int32_t)c->launcher->ths.size() : 0;
  info8[1] = g_lib_cpus_ok.load() ? (int32_t)CPU_COUNT(&g_lib_cpus) : 0;
  info8[2] = g_pinned.load() ? 1 : 0;
  info8[3] = (int32_t)std::lround(100.0 * std::min(1e6, bsr_internal_cpu_budget()));
  info8[4] = c->tile_T;
  info8[5] = c->tile_slices;
  info8[6] = c->tile_bps;
  info8[7] = c->tile_whole ? (c->tile_asm ? 3 : 1) : (c->tile_stream ? 2 : 0);   // 1: whole slices in LDS (k_tile1), 3: the same with the tape loop in assembly (k_tile1a), 2: streaming kernel, 0: chunked k_tile / k_rows
  return BSR_OK;
}

extern "C" int bsr_place_info(int32_t* info4) {
  if (!info4) return BSR_E_ARG;
  info4[0] = g_lib_cpus_ok.load() ? 1 : 0;
  info4[1] = info4[0] ? (int32_t)CPU_COUNT(&g_lib_cpus) : 0;
  info4[2] = g_lib_numa;
  info4[3] = g_pinned.load() ? 1 : 0;
  return BSR_OK;
}

extern "C" int bsr_set_profiling(bsr_ctx* c, int32_t level) {
  if (!c) return BSR_E_ARG;
  c->prof = std::max(0, std::min(2, (int)level));
  return BSR_OK;
}
extern "C" int bsr_last_timing(bsr_ctx* c, double* us5) {
  if (!c || !us5) return BSR_E_ARG;
  memcpy(us5, c->last_us, sizeof c->last_us);
  return BSR_OK;
}

