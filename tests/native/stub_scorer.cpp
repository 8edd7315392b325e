// TEST INFRASTRUCTURE -- not part of the product, never loaded by it.
//
// A CPU stand-in for the data side of libbsr_hip.so (include/bsr_hip.h), in plain double arithmetic, so that the
// threaded HOST code of the product -- the native sampler of csrc/bsr_engine.hip: worker threads, batches generated
// ahead, the context lock around the accept path -- can run under AddressSanitizer / UndefinedBehaviorSanitizer /
// ThreadSanitizer on a box without a GPU (GPU sanitizers are not available on the pool).  tests/native/build_san.sh
// compiles bsr_engine.hip as C++ (-DBSR_HOST_ONLY) together with this file; tests/test_sanitizers.py runs the golden
// traces of the reference through the result.  Semantics follow oracle/bsr_oracle.py (allcal, score_proposal,
// yloglike, intercept_fit), i.e. codes/funcs.py:175-220, 1147-1174, 1226 and codes/bsr_class.py:147-163.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bsr_hip.h"

struct bsr_ctx {
  int64_t N = 0;
  int d = 0, K = 0, n_chains = 0, max_batch = 0;
  std::vector<double> X, y;                       // row-major X
  std::vector<std::vector<double>> cur;           // [chain*K + k][N]
  struct Slot {
    std::vector<bsr_node> rows;
    std::vector<int32_t> off;
    std::vector<bsr_score> out;
    bool scored = false;
  } slot[2 * BSR_MAX_INFLIGHT];
  int next_slot = 0;
  std::vector<double> lo, hi;
  std::mutex mu;
  std::string err;
};

static thread_local std::string g_err;
static const double kInf = INFINITY, kNaN = NAN;

static int eval_tape(const bsr_ctx* c, const bsr_node* t, int len, std::vector<double>& out) {
  const int64_t N = c->N;
  std::vector<std::vector<double>> st;
  for (int i = 0; i < len; ++i) {
    const bsr_node& n = t[i];
    if (n.opcode == BSR_OP_TERMINAL) {
      if (n.feature < 0 || n.feature >= c->d) return BSR_E_TAPE;
      std::vector<double> v((size_t)N);
      for (int64_t r = 0; r < N; ++r) v[r] = c->X[(size_t)r * c->d + n.feature];
      st.push_back(std::move(v));
    } else if (n.opcode == BSR_OP_ADD || n.opcode == BSR_OP_MUL || n.opcode == BSR_OP_SUB || n.opcode == BSR_OP_DIV) {
      if (st.size() < 2) return BSR_E_TAPE;
      std::vector<double> b = std::move(st.back());
      st.pop_back();
      std::vector<double>& a = st.back();
      for (int64_t r = 0; r < N; ++r) {
        switch (n.opcode) {
          case BSR_OP_ADD: a[r] = a[r] + b[r]; break;
          case BSR_OP_MUL: a[r] = a[r] * b[r]; break;
          case BSR_OP_SUB: a[r] = a[r] - b[r]; break;
          default: a[r] = (b[r] == 0.0) ? 0.0 : a[r] / b[r]; break;
        }
      }
    } else {
      if (st.empty()) return BSR_E_TAPE;
      std::vector<double>& a = st.back();
      for (int64_t r = 0; r < N; ++r) {
        const double x = a[r];
        switch (n.opcode) {
          case BSR_OP_INV: a[r] = (x == 0.0) ? 0.0 : 1.0 / x; break;
          case BSR_OP_LN: { const double m = n.a * x; a[r] = m + n.b; } break;
          case BSR_OP_NEG: a[r] = -x; break;
          case BSR_OP_SIN: a[r] = std::sin(x); break;
          case BSR_OP_COS: a[r] = std::cos(x); break;
          case BSR_OP_EXP: a[r] = (x <= 200.0) ? std::exp(x) : 1e10; break;
          case BSR_OP_SQUARE: a[r] = x * x; break;
          case BSR_OP_CUBIC: a[r] = std::pow(x, 3.0); break;
          case BSR_OP_LOG: a[r] = (x == 0.0) ? 0.0 : std::log(std::fabs(x)); break;
          default: return BSR_E_TAPE;
        }
      }
    }
  }
  if (st.size() != 1) return BSR_E_TAPE;
  out = std::move(st.back());
  return BSR_OK;
}

// singular values of the N x K matrix of columns (one-sided Jacobi on a copy)
static void singular_values(std::vector<std::vector<double>> cols, std::vector<double>& sv) {
  const int K = (int)cols.size();
  const size_t N = K ? cols[0].size() : 0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int a = 0; a < K - 1; ++a)
      for (int b = a + 1; b < K; ++b) {
        double al = 0, be = 0, ga = 0;
        for (size_t r = 0; r < N; ++r) {
          al += cols[a][r] * cols[a][r];
          be += cols[b][r] * cols[b][r];
          ga += cols[a][r] * cols[b][r];
        }
        if (al * be > 0.0 && ga * ga > 1e-34 * al * be) {
          off = std::max(off, ga * ga / (al * be));
          const double zeta = (be - al) / (2.0 * ga);
          const double t = std::copysign(1.0, zeta) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
          for (size_t r = 0; r < N; ++r) {
            const double wa = cols[a][r], wb = cols[b][r];
            cols[a][r] = cs * wa - sn * wb;
            cols[b][r] = sn * wa + cs * wb;
          }
        }
      }
    if (off <= 1e-30) break;
  }
  sv.assign(K, 0.0);
  for (int j = 0; j < K; ++j) {
    double s = 0;
    for (size_t r = 0; r < N; ++r) s += cols[j][r] * cols[j][r];
    sv[j] = std::sqrt(s);
  }
}

// beta = inv(A^T A + 1e-6 I) A^T y on the scaled columns (Gauss-Jordan with partial pivoting); returns SSE
static double ridge_fit(const std::vector<const double*>& cols, const std::vector<double>& y, double scale,
                        std::vector<double>& beta) {
  const int K = (int)cols.size();
  const size_t N = y.size();
  std::vector<double> G((size_t)K * (K + 1), 0.0);
  for (int i = 0; i < K; ++i) {
    for (int j = 0; j < K; ++j) {
      double s = 0;
      for (size_t r = 0; r < N; ++r) s += (cols[i][r] / scale) * (cols[j][r] / scale);
      G[(size_t)i * (K + 1) + j] = s + (i == j ? 1e-6 : 0.0);
    }
    double s = 0;
    for (size_t r = 0; r < N; ++r) s += (cols[i][r] / scale) * y[r];
    G[(size_t)i * (K + 1) + K] = s;
  }
  for (int p = 0; p < K; ++p) {
    int best = p;
    for (int r = p + 1; r < K; ++r)
      if (std::fabs(G[(size_t)r * (K + 1) + p]) > std::fabs(G[(size_t)best * (K + 1) + p])) best = r;
    if (best != p)
      for (int j = 0; j <= K; ++j) std::swap(G[(size_t)p * (K + 1) + j], G[(size_t)best * (K + 1) + j]);
    const double piv = G[(size_t)p * (K + 1) + p];
    for (int j = 0; j <= K; ++j) G[(size_t)p * (K + 1) + j] /= piv;
    for (int r = 0; r < K; ++r) {
      if (r == p) continue;
      const double f = G[(size_t)r * (K + 1) + p];
      for (int j = 0; j <= K; ++j) G[(size_t)r * (K + 1) + j] -= f * G[(size_t)p * (K + 1) + j];
    }
  }
  beta.assign(K, 0.0);
  for (int i = 0; i < K; ++i) beta[i] = G[(size_t)i * (K + 1) + K];
  double sse = 0;
  for (size_t r = 0; r < N; ++r) {
    double f = 0;
    for (int i = 0; i < K; ++i) f += (cols[i][r] / scale) * beta[i];
    sse += (y[r] - f) * (y[r] - f);
  }
  return sse;
}

static void column_census(const std::vector<double>& v, double* maxabs, uint32_t* flags) {
  double m = 0;
  uint32_t f = 0;
  for (double x : v) {
    if (std::isnan(x)) f |= BSR_F_NAN;
    else if (std::isinf(x)) f |= BSR_F_INF;
    else m = std::max(m, std::fabs(x));
  }
  *maxabs = (f & BSR_F_INF) ? kInf : m;
  *flags = f;
}

static void score_one(const bsr_ctx* c, int chain, int k, const std::vector<double>& z, double sigma, bsr_score* out) {
  const int K = c->K;
  memset(out, 0, sizeof *out);
  std::vector<const double*> cols(K);
  uint32_t flags = 0;
  double scale = 0;
  for (int j = 0; j < K; ++j) {
    const std::vector<double>& col = (j == k) ? z : c->cur[(size_t)chain * K + j];
    cols[j] = col.data();
    double m;
    uint32_t f;
    column_census(col, &m, &f);
    flags |= f;
    scale = std::max(scale, m);
    if (j == k) out->maxabs = m;
  }
  out->flags = flags;
  out->scale = scale;
  if (flags & BSR_F_NAN) { out->rank = -1; out->loglik = out->sse = kNaN; out->flags |= BSR_F_RANKDEF; return; }
  if (flags & BSR_F_INF) { out->rank = 0; out->loglik = out->sse = kNaN; out->flags |= BSR_F_RANKDEF; return; }
  std::vector<std::vector<double>> M(K);
  for (int j = 0; j < K; ++j) M[j].assign(cols[j], cols[j] + c->N);
  std::vector<double> sv;
  singular_values(M, sv);
  const double smax = *std::max_element(sv.begin(), sv.end());
  const double tol = smax * (double)std::max<int64_t>(c->N, K) * 2.220446049250313e-16;
  int rank = 0;
  for (double s : sv) rank += (s > tol) ? 1 : 0;
  out->rank = rank;
  out->smax = smax;
  out->smin = *std::min_element(sv.begin(), sv.end());
  if (rank < K) { out->flags |= BSR_F_RANKDEF; out->loglik = out->sse = kNaN; return; }
  std::vector<double> beta;
  const double sse = ridge_fit(cols, c->y, scale, beta);
  out->sse = sse;
  out->loglik = -sse / (2 * sigma * sigma) - 0.5 * (double)c->N * std::log(2 * M_PI * sigma * sigma);
  for (int j = 0; j < K; ++j) out->beta[j] = beta[j];
}

static int do_submit(bsr_ctx* c, int si, const bsr_node* rows, const int32_t* off, const int32_t* chain,
                     const int32_t* which_k, const double* sigma, int32_t B) {
  if (!c || !rows || !off || B <= 0 || B > c->max_batch) return BSR_E_ARG;
  bsr_ctx::Slot& s = c->slot[si];
  s.rows.assign(rows, rows + off[B]);
  s.off.assign(off, off + B + 1);
  s.out.resize((size_t)B);
  for (int i = 0; i < B; ++i) {
    std::vector<double> z;
    const int rc = eval_tape(c, rows + off[i], off[i + 1] - off[i], z);
    if (rc != BSR_OK) return rc;
    score_one(c, chain[i], which_k[i], z, sigma[i], &s.out[i]);
  }
  s.scored = true;
  return BSR_OK;
}

extern "C" {
int bsr_abi_version(void) { return BSR_ABI_VERSION; }
int bsr_device_count(int* n) { if (n) *n = 1; return BSR_OK; }
const char* bsr_last_error(const bsr_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }
int bsr_ctx_create_tuned(bsr_ctx** out, int dev, int64_t N, int32_t d, const double* X, const double* y, int32_t K,
                         int32_t n_chains, int32_t max_batch, int32_t dtype, int32_t, int32_t) {
  return bsr_ctx_create(out, dev, N, d, X, y, K, n_chains, max_batch, dtype);
}
int bsr_ctx_create(bsr_ctx** out, int, int64_t N, int32_t d, const double* X, const double* y, int32_t K,
                   int32_t n_chains, int32_t max_batch, int32_t) {
  if (!out || !X || N <= 0 || d <= 0 || K < 0 || K > BSR_MAX_K) return BSR_E_ARG;
  bsr_ctx* c = new bsr_ctx();
  c->N = N; c->d = d; c->K = K; c->n_chains = n_chains; c->max_batch = max_batch;
  c->X.assign(X, X + (size_t)N * d);
  if (y) c->y.assign(y, y + N);
  c->cur.assign((size_t)std::max(1, n_chains) * std::max(1, K), std::vector<double>((size_t)N, 0.0));
  c->lo.assign(d, kInf);
  c->hi.assign(d, -kInf);
  for (int64_t r = 0; r < N; ++r)
    for (int f = 0; f < d; ++f) {
      c->lo[f] = std::min(c->lo[f], X[(size_t)r * d + f]);
      c->hi[f] = std::max(c->hi[f], X[(size_t)r * d + f]);
    }
  *out = c;
  return BSR_OK;
}
int bsr_ctx_destroy(bsr_ctx* c) { delete c; return BSR_OK; }
int bsr_eval_tapes(bsr_ctx* c, const bsr_node* rows, const int32_t* off, int32_t n, double* out_cols, double* maxabs,
                   uint32_t* flags) {
  if (!c) return BSR_E_ARG;
  for (int i = 0; i < n; ++i) {
    std::vector<double> z;
    const int rc = eval_tape(c, rows + off[i], off[i + 1] - off[i], z);
    if (rc != BSR_OK) return rc;
    double m;
    uint32_t f;
    column_census(z, &m, &f);
    if (out_cols) memcpy(out_cols + (size_t)i * c->N, z.data(), sizeof(double) * c->N);
    if (maxabs) maxabs[i] = m;
    if (flags) flags[i] = f;
  }
  return BSR_OK;
}
int bsr_set_current(bsr_ctx* c, int32_t chain, int32_t k, const bsr_node* tape, int32_t len) {
  if (!c || chain < 0 || chain >= c->n_chains || k < 0 || k >= c->K) return BSR_E_ARG;
  return eval_tape(c, tape, len, c->cur[(size_t)chain * c->K + k]);
}
extern "C++" int bsr_internal_commit(bsr_ctx* c, int si, int32_t chain, int32_t k, int32_t idx) {
  bsr_ctx::Slot& s = c->slot[si];
  if (!s.scored || idx < 0 || idx + 1 >= (int)s.off.size()) return BSR_E_STATE;
  return eval_tape(c, s.rows.data() + s.off[idx], s.off[idx + 1] - s.off[idx], c->cur[(size_t)chain * c->K + k]);
}
static int g_last_waited = 0;
int bsr_commit(bsr_ctx* c, int32_t chain, int32_t k, int32_t idx) { return bsr_internal_commit(c, g_last_waited, chain, k, idx); }
int bsr_refresh(bsr_ctx* c, int32_t chain, bsr_chain_info* info) {
  if (!c || !info || chain < 0 || chain >= c->n_chains) return BSR_E_ARG;
  memset(info, 0, sizeof *info);
  const int K = c->K;
  std::vector<const double*> cols(K);
  uint32_t any = 0;
  double scale = 0;
  for (int j = 0; j < K; ++j) {
    cols[j] = c->cur[(size_t)chain * K + j].data();
    column_census(c->cur[(size_t)chain * K + j], &info->maxabs[j], &info->colflags[j]);
    any |= info->colflags[j];
    scale = std::max(scale, info->maxabs[j]);
  }
  info->scale_old = scale;
  if (any) { info->sse_old = kNaN; return BSR_OK; }
  std::vector<double> beta;
  info->sse_old = ridge_fit(cols, c->y, scale, beta);
  for (int j = 0; j < K; ++j) info->beta_old[j] = beta[j];
  return BSR_OK;
}
int bsr_fit_beta(bsr_ctx* c, int32_t chain, double* beta_out, double* rmse_out) {   // codes/bsr_class.py:147-163
  if (!c || !beta_out || !rmse_out) return BSR_E_ARG;
  const int K = c->K;
  std::vector<double> ones((size_t)c->N, 1.0);
  std::vector<const double*> cols(K + 1);
  cols[0] = ones.data();
  double scale = 1.0;
  for (int j = 0; j < K; ++j) {
    cols[j + 1] = c->cur[(size_t)chain * K + j].data();
    double m;
    uint32_t f;
    column_census(c->cur[(size_t)chain * K + j], &m, &f);
    scale = std::max(scale, m);
  }
  std::vector<double> beta;
  const double sse = ridge_fit(cols, c->y, scale, beta);
  for (int j = 0; j <= K; ++j) beta_out[j] = beta[j] / scale;
  *rmse_out = std::sqrt(sse / (double)c->N);
  return BSR_OK;
}
int bsr_get_current(bsr_ctx* c, int32_t chain, double* out) {
  for (int k = 0; k < c->K; ++k) memcpy(out + (size_t)k * c->N, c->cur[(size_t)chain * c->K + k].data(), sizeof(double) * c->N);
  return BSR_OK;
}
extern "C++" int bsr_internal_submit(bsr_ctx* c, int si, const bsr_node* rows, const int32_t* off, const int32_t* chain,
                        const int32_t* which_k, const double* sigma, int32_t B) {
  return do_submit(c, si, rows, off, chain, which_k, sigma, B);
}
extern "C++" int bsr_internal_wait(bsr_ctx* c, int si, bsr_score* out) {
  bsr_ctx::Slot& s = c->slot[si];
  if (!s.scored) return BSR_E_STATE;
  memcpy(out, s.out.data(), sizeof(bsr_score) * s.out.size());
  return BSR_OK;
}
int bsr_score_submit(bsr_ctx* c, const bsr_node* rows, const int32_t* off, const int32_t* chain, const int32_t* which_k,
                     const double* sigma, int32_t B, int32_t* ticket) {
  const int si = c->next_slot;
  const int rc = do_submit(c, si, rows, off, chain, which_k, sigma, B);
  if (rc != BSR_OK) return rc;
  *ticket = si;
  c->next_slot = (si + 1) % BSR_MAX_INFLIGHT;
  return BSR_OK;
}
int bsr_score_wait(bsr_ctx* c, int32_t ticket, bsr_score* out) {
  g_last_waited = ticket;
  return bsr_internal_wait(c, ticket, out);
}
int bsr_score_batch(bsr_ctx* c, const bsr_node* rows, const int32_t* off, const int32_t* chain, const int32_t* which_k,
                    const double* sigma, int32_t B, bsr_score* out) {
  int32_t t = -1;
  const int rc = bsr_score_submit(c, rows, off, chain, which_k, sigma, B, &t);
  return rc != BSR_OK ? rc : bsr_score_wait(c, t, out);
}
// the device-side MH step is a GPU feature: the stub does not offer it (the sampler's default does not use it)
int bsr_score_submit_mh(bsr_ctx*, const bsr_node*, const int32_t*, const int32_t*, const int32_t*, const double*, int32_t,
                        const double*, const int32_t*, const int32_t*, int32_t, int32_t*) { return BSR_E_STATE; }
int bsr_score_wait_mh(bsr_ctx*, int32_t, bsr_score*, bsr_event*) { return BSR_E_STATE; }
// (the device-side MH step is a GPU feature: with spans the stub refuses; without, this is the plain submit -- the sampler
// passes its lone chain group's batches this way, with staging deferred)
extern "C++" int bsr_internal_submit_mh(bsr_ctx* c, int si, const bsr_node* rows, const int32_t* off, const int32_t* chain,
                           const int32_t* which_k, const double* sigma, int32_t B, const double*, const int32_t*,
                           const int32_t*, int32_t n_spans, bool) {
  if (n_spans > 0) return BSR_E_STATE;
  return do_submit(c, si, rows, off, chain, which_k, sigma, B);
}
extern "C++" int bsr_internal_wait_mh(bsr_ctx*, int, bsr_score*, bsr_event*) { return BSR_E_STATE; }
int bsr_yloglike_host(int, int64_t, int32_t, const double*, const double*, double, int32_t, double*, double*, double*,
                      double*, int32_t*) { return BSR_E_NODEVICE; }
int bsr_set_profiling(bsr_ctx*, int32_t) { return BSR_OK; }
int bsr_last_timing(bsr_ctx*, double* us5) { for (int i = 0; i < 5; ++i) us5[i] = 0; return BSR_OK; }
int bsr_ctx_info(const bsr_ctx*, int32_t* v) { for (int i = 0; i < 8; ++i) v[i] = 0; return BSR_OK; }
int bsr_batch_stats(const bsr_ctx*, int32_t, int32_t* v) { for (int i = 0; i < 4; ++i) v[i] = 0; return BSR_OK; }
int bsr_dispatch_info(const bsr_ctx*, int64_t* v) { for (int i = 0; i < 8; ++i) v[i] = 0; return BSR_OK; }
int bsr_place_info(int32_t* v) { for (int i = 0; i < 4; ++i) v[i] = -1; return BSR_OK; }
int bsr_comm_unique_id(void*) { return BSR_E_COMM; }
int bsr_comm_init(bsr_ctx*, int32_t, int32_t, const void*) { return BSR_E_COMM; }
int bsr_comm_allgather(bsr_ctx*, const void*, void*, int64_t) { return BSR_E_COMM; }
int bsr_comm_destroy(bsr_ctx*) { return BSR_OK; }
}  // extern "C"

void bsr_internal_lock(bsr_ctx* c) { c->mu.lock(); }
void bsr_internal_unlock(bsr_ctx* c) { c->mu.unlock(); }
void bsr_internal_feature_range(const bsr_ctx* c, const double** lo, const double** hi) { *lo = c->lo.data(); *hi = c->hi.data(); }
double bsr_internal_cpu_budget() { return 16.0; }
void bsr_internal_place_thread() {}
int bsr_internal_placed_cpus() { return 0; }
