// C ABI of libbsr_hip.so (include/bsr_hip.h): context, chain cache, batch scoring, RCCL gather.
// Host-side orchestration only; all O(N) work is in bsr_kernels.hip.
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bsr_internal.h"

static thread_local std::string g_create_error;

struct bsr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t N = 0, ld = 0;
  int d = 0, K = 0, n_chains = 0, max_batch = 0, dtype = 0;
  size_t esz = 8;
  bool has_y = false;
  void* Xt = nullptr;
  void* y = nullptr;
  void* cur = nullptr;   // [chain][k][ld]
  void* Q = nullptr;     // [chain][k][K-1][ld]
  void* zbuf = nullptr;  // [max_batch][ld]
  ChainK* d_ck = nullptr;
  std::vector<ChainK> h_ck;
  ChainFitOut* d_fit = nullptr;  // [chain] no-intercept fit (also carries per-column max/flags) + 1 scratch slot
  std::vector<ChainFitOut> h_fit;
  std::vector<char> ready;       // chain factors valid
  std::vector<char> col_set;     // [chain*K+k] column initialised
  // per-batch buffers
  bsr_node* d_tapes = nullptr;
  bsr_node* h_tapes = nullptr;
  size_t tapes_cap = 0;
  PropDesc* d_desc = nullptr;
  PropDesc* h_desc = nullptr;
  PropCoef* d_coef = nullptr;
  bsr_score* d_out = nullptr;
  bsr_score* h_out = nullptr;
  double* part1 = nullptr;
  double* part2 = nullptr;
  size_t part_cap = 0;  // in (proposal,row block) records
  double* spill = nullptr;
  size_t spill_cap = 0;  // bytes
  double* d_stage = nullptr;  // fp64 staging for column download in f32 mode
  int last_B = 0;
  // tuning
  int rb_rows = 512;
  int target_wgs = 2048;
  // profiling
  bool prof = false;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double last_us[5] = {0, 0, 0, 0, 0};
  ncclComm_t comm = nullptr;
  void* comm_buf = nullptr;
  size_t comm_cap = 0;
  std::string err;
};

#define HIPCHK(ctx, call)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      char b_[512];                                                                             \
      snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (ctx)->err = b_;                                                                          \
      return BSR_E_HIP;                                                                         \
    }                                                                                           \
  } while (0)

static int fail(bsr_ctx* ctx, int code, const char* msg) {
  if (ctx) ctx->err = msg; else g_create_error = msg;
  return code;
}

static inline void* col_ptr(const bsr_ctx* c, void* base, int64_t col) {
  return (char*)base + (size_t)col * c->ld * c->esz;
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

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

extern "C" int bsr_ctx_destroy(bsr_ctx* c) {
  if (!c) return BSR_E_ARG;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
  void* dev[] = {c->Xt, c->y, c->cur, c->Q, c->zbuf, c->d_ck, c->d_fit, c->d_tapes, c->d_desc, c->d_coef,
                 c->d_out, c->part1, c->part2, c->spill, c->d_stage, c->comm_buf};
  for (void* p : dev) if (p) (void)hipFree(p);
  if (c->h_tapes) (void)hipHostFree(c->h_tapes);
  if (c->h_desc) (void)hipHostFree(c->h_desc);
  if (c->h_out) (void)hipHostFree(c->h_out);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
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
  if (!out) return BSR_E_ARG;
  *out = nullptr;
  if (!X || N <= 0 || d <= 0 || d > 64 * 1024) return fail(nullptr, BSR_E_ARG, "bsr_ctx_create: bad X/N/d");
  if (K < 0 || K > BSR_MAX_K || n_chains < 0 || max_batch <= 0)
    return fail(nullptr, BSR_E_ARG, "bsr_ctx_create: bad K/n_chains/max_batch");
  if (dtype != BSR_DTYPE_F64 && dtype != BSR_DTYPE_F32) return fail(nullptr, BSR_E_ARG, "bsr_ctx_create: bad dtype");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
    return fail(nullptr, BSR_E_NODEVICE, "bsr_ctx_create: no such HIP device");
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
  c->rb_rows = env_int("BSR_RB_ROWS", 512);
  if (c->rb_rows < 128 || c->rb_rows > BSR_ROW_ALIGN || (BSR_ROW_ALIGN % c->rb_rows) != 0) c->rb_rows = 512;
  c->target_wgs = env_int("BSR_TARGET_WGS", 2048);
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
  CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  const size_t colb = (size_t)c->ld * c->esz;
  CK(hipMalloc(&c->Xt, colb * d));
  CK(hipMemsetAsync(c->Xt, 0, colb * d, c->stream));
  CK(hipMalloc(&c->y, colb));
  CK(hipMemsetAsync(c->y, 0, colb, c->stream));
  CK(hipMalloc(&c->zbuf, colb * max_batch));
  CK(hipMemsetAsync(c->zbuf, 0, colb * max_batch, c->stream));
  if (K > 0 && n_chains > 0) {
    CK(hipMalloc(&c->cur, colb * n_chains * K));
    CK(hipMemsetAsync(c->cur, 0, colb * n_chains * K, c->stream));
    if (K > 1) {
      CK(hipMalloc(&c->Q, colb * n_chains * K * (K - 1)));
      CK(hipMemsetAsync(c->Q, 0, colb * n_chains * K * (K - 1), c->stream));
    }
    CK(hipMalloc((void**)&c->d_ck, sizeof(ChainK) * n_chains * K));
    CK(hipMemsetAsync(c->d_ck, 0, sizeof(ChainK) * n_chains * K, c->stream));
    c->h_ck.resize((size_t)n_chains * K);
    c->ready.assign(n_chains, 0);
    c->col_set.assign((size_t)n_chains * K, 0);
  }
  CK(hipMalloc((void**)&c->d_fit, sizeof(ChainFitOut) * (n_chains + 1)));
  c->h_fit.resize(n_chains + 1);
  CK(hipMalloc((void**)&c->d_desc, sizeof(PropDesc) * max_batch));
  CK(hipHostMalloc((void**)&c->h_desc, sizeof(PropDesc) * max_batch));
  CK(hipMalloc((void**)&c->d_coef, sizeof(PropCoef) * max_batch));
  CK(hipMalloc((void**)&c->d_out, sizeof(bsr_score) * max_batch));
  CK(hipHostMalloc((void**)&c->h_out, sizeof(bsr_score) * max_batch));
  for (auto& e : c->ev) CK(hipEventCreate(&e));
#undef CK
  rc = (dtype == BSR_DTYPE_F64) ? upload_data<double>(c, X, y) : upload_data<float>(c, X, y);
  if (rc != BSR_OK) return bail(rc);
  *out = c;
  return BSR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
static int check_tape(bsr_ctx* c, const bsr_node* t, int len, int* max_sp) {
  if (len <= 0) return fail(c, BSR_E_TAPE, "empty tape");
  if (len > BSR_MAX_TAPE) return fail(c, BSR_E_TOOBIG, "tape longer than BSR_MAX_TAPE");
  int sp = 0, mx = 0;
  for (int i = 0; i < len; ++i) {
    const int op = t[i].opcode;
    if (op == BSR_OP_TERMINAL) {
      if (t[i].feature < 0 || t[i].feature >= c->d) return fail(c, BSR_E_TAPE, "terminal feature out of range");
      ++sp;
    } else if (op >= 0 && op < BSR_OP_ADD) {
      if (sp < 1) return fail(c, BSR_E_TAPE, "unary operator on empty stack");
    } else if (op == BSR_OP_ADD || op == BSR_OP_MUL) {
      if (sp < 2) return fail(c, BSR_E_TAPE, "binary operator needs two operands");
      --sp;
    } else {
      return fail(c, BSR_E_TAPE, "unknown opcode");
    }
    mx = std::max(mx, sp);
  }
  if (sp != 1) return fail(c, BSR_E_TAPE, "tape does not reduce to one value");
  if (mx > BSR_MAX_STACK) return fail(c, BSR_E_TOOBIG, "tape needs a deeper stack than BSR_MAX_STACK");
  *max_sp = mx;
  return BSR_OK;
}

static int ensure_tapes(bsr_ctx* c, size_t rows) {
  if (rows <= c->tapes_cap) return BSR_OK;
  size_t cap = std::max(rows, std::max((size_t)4096, c->tapes_cap * 2));
  if (c->d_tapes) HIPCHK(c, hipFree(c->d_tapes));
  if (c->h_tapes) HIPCHK(c, hipHostFree(c->h_tapes));
  c->d_tapes = nullptr;
  c->h_tapes = nullptr;
  HIPCHK(c, hipMalloc((void**)&c->d_tapes, cap * sizeof(bsr_node)));
  HIPCHK(c, hipHostMalloc((void**)&c->h_tapes, cap * sizeof(bsr_node)));
  c->tapes_cap = cap;
  return BSR_OK;
}

static LaunchGeom geometry(const bsr_ctx* c, int P) {
  LaunchGeom g;
  g.rb_rows = c->rb_rows;
  g.n_rb = (int)((c->N + g.rb_rows - 1) / g.rb_rows);
  const int want_pg = std::max(1, c->target_wgs / g.n_rb);       // proposal groups wanted
  int pg = (P + want_pg - 1) / want_pg;                            // proposals per workgroup
  if (P >= BSR_WG_WAVES) pg = (pg + BSR_WG_WAVES - 1) / BSR_WG_WAVES * BSR_WG_WAVES;
  pg = std::max(1, std::min(pg, P));
  g.pg = pg;
  g.n_pg = (P + pg - 1) / pg;
  return g;
}

static int ensure_partials(bsr_ctx* c, const LaunchGeom& g, int P, int spill_slots) {
  const size_t recs = (size_t)P * g.n_rb;
  if (recs > c->part_cap) {
    if (c->part1) HIPCHK(c, hipFree(c->part1));
    if (c->part2) HIPCHK(c, hipFree(c->part2));
    c->part1 = c->part2 = nullptr;
    const size_t cap = recs + recs / 2;
    HIPCHK(c, hipMalloc((void**)&c->part1, cap * BSR_P1_WORDS * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&c->part2, cap * BSR_P2_WORDS * sizeof(double)));
    c->part_cap = cap;
  }
  if (spill_slots > 0) {
    const size_t need = (size_t)g.n_rb * g.n_pg * BSR_WG_WAVES * spill_slots * BSR_WAVE * 2 * c->esz;
    if (need > c->spill_cap) {
      if (c->spill) HIPCHK(c, hipFree(c->spill));
      c->spill = nullptr;
      HIPCHK(c, hipMalloc((void**)&c->spill, need));
      c->spill_cap = need;
    }
  }
  return BSR_OK;
}

// Runs the four kernels over the P descriptors already in h_desc (tapes already uploaded); results land in h_out.
static int run_descs(bsr_ctx* c, int P, bool need_pass2) {
  int spill_slots = 0;
  for (int i = 0; i < P; ++i) spill_slots = std::max(spill_slots, c->h_desc[i].spill_need);
  const LaunchGeom g = geometry(c, P);
  int rc = ensure_partials(c, g, P, spill_slots);
  if (rc != BSR_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(PropDesc) * P, hipMemcpyHostToDevice, c->stream));
  const bool f64 = (c->dtype == BSR_DTYPE_F64);
  if (c->prof) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  if (f64)
    launch_pass1<double>(c->stream, g, (const double*)c->Xt, c->has_y ? (const double*)c->y : nullptr, c->ld, c->N,
                         c->d_tapes, c->d_desc, P, c->part1, spill_slots ? c->spill : nullptr, spill_slots);
  else
    launch_pass1<float>(c->stream, g, (const float*)c->Xt, c->has_y ? (const float*)c->y : nullptr, c->ld, c->N,
                        c->d_tapes, c->d_desc, P, c->part1, spill_slots ? c->spill : nullptr, spill_slots);
  if (c->prof) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  launch_solve(c->stream, c->d_desc, c->d_ck, P, g.n_rb, c->part1, c->N, c->d_coef, c->d_out);
  if (c->prof) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  if (need_pass2) {
    if (f64)
      launch_pass2<double>(c->stream, g, (const double*)c->y, c->ld, c->N, c->d_desc, c->d_coef, P, c->part2);
    else
      launch_pass2<float>(c->stream, g, (const float*)c->y, c->ld, c->N, c->d_desc, c->d_coef, P, c->part2);
  }
  if (c->prof) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
  if (need_pass2) launch_finalize(c->stream, c->d_desc, c->d_ck, c->d_coef, P, g.n_rb, c->part2, c->N, c->d_out);
  if (c->prof) HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, sizeof(bsr_score) * P, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  if (c->prof) {
    float ms = 0;
    for (int i = 0; i < 4; ++i) {
      HIPCHK(c, hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
      c->last_us[i] = ms * 1e3;
    }
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[4]));
    c->last_us[4] = ms * 1e3;
  }
  return BSR_OK;
}

static int stage_tapes(bsr_ctx* c, const bsr_node* rows, const int32_t* tape_off, int n, std::vector<int>* max_sp) {
  if (!rows || !tape_off || n <= 0) return fail(c, BSR_E_ARG, "null tapes / empty batch");
  if (n > c->max_batch) return fail(c, BSR_E_TOOBIG, "batch larger than max_batch");
  if (tape_off[0] != 0) return fail(c, BSR_E_ARG, "tape_off[0] must be 0");
  max_sp->resize(n);
  for (int i = 0; i < n; ++i) {
    const int len = tape_off[i + 1] - tape_off[i];
    int rc = check_tape(c, rows + tape_off[i], len, &(*max_sp)[i]);
    if (rc != BSR_OK) return rc;
  }
  const size_t total = (size_t)tape_off[n];
  int rc = ensure_tapes(c, total);
  if (rc != BSR_OK) return rc;
  memcpy(c->h_tapes, rows, total * sizeof(bsr_node));
  HIPCHK(c, hipMemcpyAsync(c->d_tapes, c->h_tapes, total * sizeof(bsr_node), hipMemcpyHostToDevice, c->stream));
  return BSR_OK;
}

static void fill_eval_desc(bsr_ctx* c, PropDesc* D, int off, int len, int max_sp, void* zout) {
  memset(D, 0, sizeof *D);
  D->tape_off = off;
  D->tape_len = len;
  D->mode = BSR_MODE_EVAL;
  D->nq = 0;
  D->K = c->K;
  D->spill_need = std::max(0, max_sp - 1 - BSR_REG_STACK);
  D->qbase = nullptr;
  D->zout = zout;
  D->s = 1.0;
  D->sigma = 1.0;
}

extern "C" int bsr_eval_tapes(bsr_ctx* c, const bsr_node* rows, const int32_t* tape_off, int32_t n_tapes,
                              double* out_cols, double* maxabs, uint32_t* flags) {
  if (!c) return BSR_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<int> msp;
  int rc = stage_tapes(c, rows, tape_off, n_tapes, &msp);
  if (rc != BSR_OK) return rc;
  for (int i = 0; i < n_tapes; ++i)
    fill_eval_desc(c, &c->h_desc[i], tape_off[i], tape_off[i + 1] - tape_off[i], msp[i], col_ptr(c, c->zbuf, i));
  rc = run_descs(c, n_tapes, false);
  if (rc != BSR_OK) return rc;
  c->last_B = 0;  // candidate slots no longer hold a scored batch
  for (int i = 0; i < n_tapes; ++i) {
    if (maxabs) maxabs[i] = c->h_out[i].maxabs;
    if (flags) flags[i] = c->h_out[i].flags & (BSR_F_INF | BSR_F_NAN);
  }
  if (out_cols) {
    if (c->dtype == BSR_DTYPE_F64) {
      HIPCHK(c, hipMemcpy2DAsync(out_cols, (size_t)c->N * 8, c->zbuf, (size_t)c->ld * 8, (size_t)c->N * 8, n_tapes,
                                 hipMemcpyDeviceToHost, c->stream));
    } else {
      if (!c->d_stage) HIPCHK(c, hipMalloc((void**)&c->d_stage, (size_t)c->ld * 8));
      for (int i = 0; i < n_tapes; ++i) {
        launch_convert_out<float>(c->stream, (const float*)col_ptr(c, c->zbuf, i), c->d_stage, c->N);
        HIPCHK(c, hipMemcpyAsync(out_cols + (size_t)i * c->N, c->d_stage, (size_t)c->N * 8, hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
      }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return BSR_OK;
}

static int chain_ok(bsr_ctx* c, int chain, int k) {
  if (chain < 0 || chain >= c->n_chains) return fail(c, BSR_E_ARG, "chain index out of range");
  if (k < 0 || k >= c->K) return fail(c, BSR_E_ARG, "tree index out of range");
  return BSR_OK;
}

extern "C" int bsr_set_current(bsr_ctx* c, int32_t chain, int32_t k, const bsr_node* tape, int32_t len) {
  if (!c || !tape) return BSR_E_ARG;
  int rc = chain_ok(c, chain, k);
  if (rc != BSR_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const int32_t off[2] = {0, len};
  std::vector<int> msp;
  rc = stage_tapes(c, tape, off, 1, &msp);
  if (rc != BSR_OK) return rc;
  fill_eval_desc(c, &c->h_desc[0], 0, len, msp[0], col_ptr(c, c->cur, (int64_t)chain * c->K + k));
  rc = run_descs(c, 1, false);
  if (rc != BSR_OK) return rc;
  c->ready[chain] = 0;
  c->col_set[(size_t)chain * c->K + k] = 1;
  return BSR_OK;
}

extern "C" int bsr_commit(bsr_ctx* c, int32_t chain, int32_t k, int32_t slot) {
  if (!c) return BSR_E_ARG;
  int rc = chain_ok(c, chain, k);
  if (rc != BSR_OK) return rc;
  if (slot < 0 || slot >= c->last_B) return fail(c, BSR_E_STATE, "bsr_commit: slot is not part of the last scored batch");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(col_ptr(c, c->cur, (int64_t)chain * c->K + k), col_ptr(c, c->zbuf, slot),
                           (size_t)c->ld * c->esz, hipMemcpyDeviceToDevice, c->stream));
  c->ready[chain] = 0;
  c->col_set[(size_t)chain * c->K + k] = 1;
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
  void* cols = col_ptr(c, c->cur, (int64_t)chain * K);
  ChainK* dck = c->d_ck + (size_t)chain * K;
  if (c->dtype == BSR_DTYPE_F64) {
    launch_chain_fit<double>(c->stream, (const double*)cols, (const double*)c->y, c->ld, c->N, K, 0, dfit);
    launch_refresh_basis<double>(c->stream, (const double*)cols,
                                 K > 1 ? (double*)col_ptr(c, c->Q, (int64_t)chain * K * (K - 1)) : nullptr,
                                 (const double*)c->y, c->ld, c->N, K, dfit->maxabs, dfit->colflags, dck);
  } else {
    launch_chain_fit<float>(c->stream, (const float*)cols, (const float*)c->y, c->ld, c->N, K, 0, dfit);
    launch_refresh_basis<float>(c->stream, (const float*)cols,
                                K > 1 ? (float*)col_ptr(c, c->Q, (int64_t)chain * K * (K - 1)) : nullptr,
                                (const float*)c->y, c->ld, c->N, K, dfit->maxabs, dfit->colflags, dck);
  }
  HIPCHK(c, hipMemcpyAsync(&c->h_fit[chain], dfit, sizeof(ChainFitOut), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&c->h_ck[(size_t)chain * K], dck, sizeof(ChainK) * K, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  c->ready[chain] = 1;
  if (info) {
    const ChainFitOut& f = c->h_fit[chain];
    memset(info, 0, sizeof *info);
    info->sse_old = f.sse;
    info->scale_old = f.scale;
    for (int k = 0; k < K; ++k) {
      info->maxabs[k] = f.maxabs[k];
      info->beta_old[k] = f.beta[k];
      info->colflags[k] = f.colflags[k];
    }
    info->rank_old = -2;
  }
  return BSR_OK;
}

extern "C" int bsr_score_batch(bsr_ctx* c, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                               const int32_t* which_k, const double* sigma, int32_t B, bsr_score* out) {
  if (!c || !chain || !which_k || !sigma || !out) return BSR_E_ARG;
  if (!c->has_y || c->K <= 0) return fail(c, BSR_E_STATE, "bsr_score_batch: context has no y / no chains");
  HIPCHK(c, hipSetDevice(c->device));
  const int K = c->K;
  for (int i = 0; i < B; ++i) {
    int rc = chain_ok(c, chain[i], which_k[i]);
    if (rc != BSR_OK) return rc;
    if (!c->ready[chain[i]]) return fail(c, BSR_E_STATE, "bsr_score_batch: chain not refreshed");
  }
  std::vector<int> msp;
  int rc = stage_tapes(c, rows, tape_off, B, &msp);
  if (rc != BSR_OK) return rc;
  for (int i = 0; i < B; ++i) {
    PropDesc* D = &c->h_desc[i];
    memset(D, 0, sizeof *D);
    const int ckidx = chain[i] * K + which_k[i];
    D->tape_off = tape_off[i];
    D->tape_len = tape_off[i + 1] - tape_off[i];
    D->mode = BSR_MODE_SCORE;
    D->nq = K - 1;
    D->k = which_k[i];
    D->K = K;
    D->ck = ckidx;
    D->spill_need = std::max(0, msp[i] - 1 - BSR_REG_STACK);
    D->qbase = (K > 1) ? col_ptr(c, c->Q, (int64_t)ckidx * (K - 1)) : nullptr;
    D->zout = col_ptr(c, c->zbuf, i);
    D->s = c->h_ck[ckidx].s;
    D->sigma = sigma[i];
  }
  rc = run_descs(c, B, true);
  if (rc != BSR_OK) return rc;
  memcpy(out, c->h_out, sizeof(bsr_score) * B);
  c->last_B = B;
  if (K == 1) {
    // no sibling column fixes the accumulation scale: rescore candidates whose |z|^2 left the double range
    std::vector<int> redo;
    for (int i = 0; i < B; ++i)
      if (out[i].flags & BSR_F_SCALE_RETRY) redo.push_back(i);
    if (!redo.empty()) {
      std::vector<PropDesc> keep(c->h_desc, c->h_desc + B);
      double timing[5];
      memcpy(timing, c->last_us, sizeof timing);
      for (size_t j = 0; j < redo.size(); ++j) {
        PropDesc D = keep[redo[j]];
        int e = 0;
        std::frexp(out[redo[j]].maxabs, &e);
        e = std::max(-1000, std::min(1000, e));
        D.s = std::ldexp(1.0, -e);
        c->h_desc[j] = D;
      }
      rc = run_descs(c, (int)redo.size(), true);
      if (rc != BSR_OK) return rc;
      for (size_t j = 0; j < redo.size(); ++j) {
        out[redo[j]] = c->h_out[j];
        out[redo[j]].flags &= ~BSR_F_SCALE_RETRY;
      }
      memcpy(c->last_us, timing, sizeof timing);
    }
  }
  return BSR_OK;
}

extern "C" int bsr_fit_beta(bsr_ctx* c, int32_t chain, double* beta_out, double* rmse_out) {
  if (!c || !beta_out || !rmse_out) return BSR_E_ARG;
  int rc = chain_ok(c, chain, 0);
  if (rc != BSR_OK) return rc;
  if (!c->has_y) return fail(c, BSR_E_STATE, "bsr_fit_beta: context has no y");
  for (int k = 0; k < c->K; ++k)
    if (!c->col_set[(size_t)chain * c->K + k]) return fail(c, BSR_E_STATE, "bsr_fit_beta: a current column was never set");
  HIPCHK(c, hipSetDevice(c->device));
  ChainFitOut* dfit = c->d_fit + c->n_chains;  // scratch slot
  void* cols = col_ptr(c, c->cur, (int64_t)chain * c->K);
  if (c->dtype == BSR_DTYPE_F64)
    launch_chain_fit<double>(c->stream, (const double*)cols, (const double*)c->y, c->ld, c->N, c->K, 1, dfit);
  else
    launch_chain_fit<float>(c->stream, (const float*)cols, (const float*)c->y, c->ld, c->N, c->K, 1, dfit);
  ChainFitOut& h = c->h_fit[c->n_chains];
  HIPCHK(c, hipMemcpyAsync(&h, dfit, sizeof(ChainFitOut), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
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

extern "C" int bsr_set_profiling(bsr_ctx* c, int32_t enable) {
  if (!c) return BSR_E_ARG;
  c->prof = enable != 0;
  return BSR_OK;
}
extern "C" int bsr_last_timing(bsr_ctx* c, double* us5) {
  if (!c || !us5) return BSR_E_ARG;
  memcpy(us5, c->last_us, sizeof c->last_us);
  return BSR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// RCCL: one communicator per process/GPU, one all-gather of fixed-size accepted-tree records.
extern "C" int bsr_comm_unique_id(void* id128) {
  if (!id128) return BSR_E_ARG;
  static_assert(sizeof(ncclUniqueId) <= BSR_COMM_ID_BYTES, "ncclUniqueId larger than BSR_COMM_ID_BYTES");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return BSR_E_COMM;
  memset(id128, 0, BSR_COMM_ID_BYTES);
  memcpy(id128, &id, sizeof id);
  return BSR_OK;
}

extern "C" int bsr_comm_init(bsr_ctx* c, int32_t nranks, int32_t rank, const void* id128) {
  if (!c || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) return BSR_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    c->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
    c->comm = nullptr;
    return BSR_E_COMM;
  }
  return BSR_OK;
}

extern "C" int bsr_comm_allgather(bsr_ctx* c, const void* send, void* recv, int64_t bytes_per_rank) {
  if (!c || !send || !recv || bytes_per_rank <= 0) return BSR_E_ARG;
  if (!c->comm) return fail(c, BSR_E_STATE, "bsr_comm_allgather: communicator not initialised");
  HIPCHK(c, hipSetDevice(c->device));
  int nranks = 0;
  ncclCommCount(c->comm, &nranks);
  const size_t need = (size_t)bytes_per_rank * (nranks + 1);
  if (need > c->comm_cap) {
    if (c->comm_buf) HIPCHK(c, hipFree(c->comm_buf));
    c->comm_buf = nullptr;
    HIPCHK(c, hipMalloc(&c->comm_buf, need));
    c->comm_cap = need;
  }
  char* dsend = (char*)c->comm_buf;
  char* drecv = dsend + bytes_per_rank;
  HIPCHK(c, hipMemcpyAsync(dsend, send, bytes_per_rank, hipMemcpyHostToDevice, c->stream));
  ncclResult_t r = ncclAllGather(dsend, drecv, (size_t)bytes_per_rank, ncclChar, c->comm, c->stream);
  if (r != ncclSuccess) {
    c->err = std::string("ncclAllGather: ") + ncclGetErrorString(r);
    return BSR_E_COMM;
  }
  HIPCHK(c, hipMemcpyAsync(recv, drecv, (size_t)bytes_per_rank * nranks, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BSR_OK;
}

extern "C" int bsr_comm_destroy(bsr_ctx* c) {
  if (!c) return BSR_E_ARG;
  if (c->comm) {
    ncclCommDestroy(c->comm);
    c->comm = nullptr;
  }
  return BSR_OK;
}
