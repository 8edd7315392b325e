// Chain-state refresh after an accepted proposal / initialisation (rare path, but 1.3 ms with one workgroup per
// kernel made it the end-to-end bottleneck once the sampler went native).  Multi-workgroup pipeline, 6 launches:
//
//   r1 k_rf_gram    (row blocks)  Gram of the K current columns, y and the constant, each column prescaled by its
//                                 own power of two
//   r2 k_rf_plan    (1 block)     old-state ridge OLS (codes/funcs.py:1148-1155) and intercept ridge OLS
//                                 (codes/bsr_class.py:147-163); Cholesky factor R1 of the Gram of the K prescaled
//                                 columns and T1 = R1^-1
//   r3 k_rf_apply1  (row blocks)  Q1 = (O D) T1, its Gram and Q1^T y; direct residual SSE of both fits
//   r4 k_rf_plan2   (1 block)     second Cholesky (CholeskyQR2): T2, R = R2 R1, Q^T y; fit results
//   r5 k_rf_apply2  (row blocks)  Q = Q1 T2 in place; |y - Q Q^T y|^2
//   r6 k_rf_final   (1 block)     reduce, publish ChainB
//
// ONE basis of all K current columns per chain (a proposal for tree k uses R with column k removed, bsr_kernels.hip
// k_solve): K basis columns to keep, read and rebuild instead of K(K-1) leave-one-out ones.
// CholeskyQR2 is as accurate as Gram-Schmidt while cond(O D) < ~1e5; a chain whose first Cholesky shows a pivot
// ratio below 1e-10 (or a non-positive pivot: dependent columns, e.g. two identical initial trees, or a column
// holding inf/NaN) is flagged and rebuilt by the single-workgroup Gram-Schmidt kernel k_refresh_basis
// (bsr_kernels.hip).
#include "bsr_internal.h"

#define RF_THREADS 256
#define RF_MAXK BSR_MAX_K
#define RF_NQ BSR_NQ_MAX
#define RF_GW 56   // Gram words per block: K(K+1)/2 + 2K + 2 <= 54

namespace {

__device__ __forceinline__ double rf_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sums NV per-thread values over the 256-thread block; result valid in thread 0 (and LDS out[])
template <int NV>
__device__ __forceinline__ void rf_block_sum(double (&v)[NV], double* sh /* 4*NV */, double* out /* NV */) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = rf_wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sh[w * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < NV) out[threadIdx.x] = ((sh[threadIdx.x] + sh[NV + threadIdx.x]) + sh[2 * NV + threadIdx.x]) + sh[3 * NV + threadIdx.x];
  __syncthreads();
}

__device__ __forceinline__ double rf_pow2_prescale(double m) {
  if (!(m > 0.0) || isinf(m)) return 1.0;
  int e;
  frexp(m, &e);
  e = max(-1000, min(1000, e));
  return ldexp(1.0, -e);
}

__device__ __forceinline__ int gidx(int i, int j, int K) {  // packed upper triangle, i <= j < K
  return i * K - (i * (i - 1)) / 2 + (j - i);
}

}  // namespace

// r1 ---------------------------------------------------------------------------------------------------------------
// K is a template parameter: the K(K+1)/2 + 2K + 2 accumulators live in registers with static indices and the rows
// are read once (the runtime-K version re-read them once per 8 words: 7 passes and a select chain per word at K = 8).
// word list = G(i,j) i<=j (packed rows), then g_i = v_i.y, then u_i = v_i.1, then yy, sy
template <typename T, int K>
__global__ __launch_bounds__(RF_THREADS) void k_rf_gram(const T* __restrict__ cols, const T* __restrict__ y, int64_t ld,
                                                        int64_t N, int rows_per_block,
                                                        const RefreshIn* __restrict__ in, double* __restrict__ part) {
  constexpr int nG = K * (K + 1) / 2;
  constexpr int nW = nG + 2 * K + 2;
  __shared__ double sh[4 * nW];
  __shared__ double outv[nW];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(N, r0 + rows_per_block);
  double sc[K];
#pragma unroll
  for (int j = 0; j < K; ++j) sc[j] = !(in->colflags[j] & (BSR_F_INF | BSR_F_NAN)) ? rf_pow2_prescale(in->colmax[j]) : 0.0;
  double acc[nW];
#pragma unroll
  for (int q = 0; q < nW; ++q) acc[q] = 0.0;
  for (int64_t n = r0 + threadIdx.x; n < r1; n += RF_THREADS) {
    double v[K];
#pragma unroll
    for (int j = 0; j < K; ++j) v[j] = (sc[j] != 0.0) ? sc[j] * (double)cols[(int64_t)j * ld + n] : 0.0;
    const double yv = (double)y[n];
    int w = 0;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int j = i; j < K; ++j, ++w) acc[w] = fma(v[i], v[j], acc[w]);
#pragma unroll
    for (int i = 0; i < K; ++i) acc[nG + i] = fma(v[i], yv, acc[nG + i]);
#pragma unroll
    for (int i = 0; i < K; ++i) acc[nG + K + i] = fma(v[i], 1.0, acc[nG + K + i]);
    acc[nG + 2 * K] = fma(yv, yv, acc[nG + 2 * K]);
    acc[nG + 2 * K + 1] = fma(yv, 1.0, acc[nG + 2 * K + 1]);
  }
  rf_block_sum<nW>(acc, sh, outv);
  if (threadIdx.x < nW) part[(size_t)blockIdx.x * RF_GW + threadIdx.x] = outv[threadIdx.x];
}

// Gauss-Jordan inverse with partial pivoting of the M x M matrix in shA (leading dimension LDM), result in shI.
// Called by the whole block; threads < LDM*LDM each own one element.
#define RF_LDM (RF_MAXK + 1)
__device__ void rf_gauss_jordan(double* shA, double* shI, int M, int* shpiv) {
  const int i = threadIdx.x / RF_LDM, j = threadIdx.x % RF_LDM;
  const bool act = threadIdx.x < RF_LDM * RF_LDM;
  if (act) shI[threadIdx.x] = (i == j) ? 1.0 : 0.0;
  __syncthreads();
  for (int col = 0; col < M; ++col) {
    if (threadIdx.x == 0) {
      double best = -1.0;
      int piv = col;
      for (int r = col; r < M; ++r) {
        const double v = fabs(shA[r * RF_LDM + col]);
        if (v > best) { best = v; piv = r; }
      }
      *shpiv = piv;
    }
    __syncthreads();
    const int piv = *shpiv;
    double a = 0.0, b = 0.0;
    if (act) {
      const int si = (i == col) ? piv : ((i == piv) ? col : i);
      const double d = shA[piv * RF_LDM + col];
      const double rowA = shA[piv * RF_LDM + j], rowI = shI[piv * RF_LDM + j];
      if (i == col) {
        a = rowA / d;
        b = rowI / d;
      } else {
        const double f = shA[si * RF_LDM + col] / d;
        a = shA[si * RF_LDM + j] - f * rowA;
        b = shI[si * RF_LDM + j] - f * rowI;
      }
    }
    __syncthreads();
    if (act) {
      shA[threadIdx.x] = a;
      shI[threadIdx.x] = b;
    }
    __syncthreads();
  }
}

// Upper Cholesky G = R^T R of the m x m matrix g (row-major ld RF_NQ) and T = R^-1, by ONE thread on its own
// LDS-resident scratch.  Returns the pivot ratio min(r_jj^2)/max(g_jj) (<= 0 on breakdown).
__device__ double rf_chol_inv(const double* g, int m, double* R, double* Tm) {
  double dmax = 0.0, pmin = INFINITY;
  for (int j = 0; j < m; ++j) dmax = fmax(dmax, g[j * RF_NQ + j]);
  for (int i = 0; i < RF_NQ * RF_NQ; ++i) {
    R[i] = 0.0;
    Tm[i] = 0.0;
  }
  for (int j = 0; j < m; ++j) {
    for (int i = 0; i <= j; ++i) {
      double sum = g[i * RF_NQ + j];
      for (int t = 0; t < i; ++t) sum -= R[t * RF_NQ + i] * R[t * RF_NQ + j];
      if (i == j) {
        pmin = fmin(pmin, sum);
        R[j * RF_NQ + j] = (sum > 0.0) ? sqrt(sum) : 0.0;
      } else {
        const double d = R[i * RF_NQ + i];
        R[i * RF_NQ + j] = (d > 0.0) ? sum / d : 0.0;
      }
    }
  }
  // T = R^-1 (upper triangular back substitution, column by column)
  for (int j = 0; j < m; ++j) {
    const double djj = R[j * RF_NQ + j];
    Tm[j * RF_NQ + j] = (djj > 0.0) ? 1.0 / djj : 0.0;
    for (int i = j - 1; i >= 0; --i) {
      double sum = 0.0;
      for (int t = i + 1; t <= j; ++t) sum += R[i * RF_NQ + t] * Tm[t * RF_NQ + j];
      const double dii = R[i * RF_NQ + i];
      Tm[i * RF_NQ + j] = (dii > 0.0) ? -sum / dii : 0.0;
    }
  }
  return (dmax > 0.0) ? pmin / dmax : 0.0;
}

// r2 ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RF_THREADS) void k_rf_plan(const RefreshIn* __restrict__ in, int K, int64_t N,
                                                        int n_blocks, const double* __restrict__ part,
                                                        RefreshPlan* __restrict__ plan) {
  __shared__ double G[RF_GW];
  __shared__ double shA[RF_LDM * RF_LDM], shI[RF_LDM * RF_LDM];
  __shared__ double sub[RF_NQ * RF_NQ], Rk[RF_NQ * RF_NQ], Tk[RF_NQ * RF_NQ];
  __shared__ double sc[RF_MAXK];
  __shared__ int shpiv;
  const int nG = K * (K + 1) / 2;
  const int nW = nG + 2 * K + 2;
  {  // cross-block reduction: 4 thread groups each take every 4th block, combined in a fixed order
    __shared__ double red[4][RF_GW];
    const int slice = threadIdx.x >> 6, w = threadIdx.x & 63;
    if (w < RF_GW) {
      double t = 0.0;
      if (w < nW) {
#pragma unroll 4
        for (int b = slice; b < n_blocks; b += 4) t += part[(size_t)b * RF_GW + w];
      }
      red[slice][w] = t;
    }
    __syncthreads();
    if (threadIdx.x < RF_GW) G[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  }
  uint32_t anyfl = 0;
  double scale = 0.0;
  for (int j = 0; j < K; ++j) {
    anyfl |= in->colflags[j] & (BSR_F_INF | BSR_F_NAN);
    scale = fmax(scale, in->colmax[j]);
  }
  if (anyfl & BSR_F_INF) scale = INFINITY;
  if (threadIdx.x < RF_MAXK)
    sc[threadIdx.x] = ((int)threadIdx.x < K && !(in->colflags[threadIdx.x] & (BSR_F_INF | BSR_F_NAN)))
                          ? rf_pow2_prescale(in->colmax[threadIdx.x]) : 0.0;
  __syncthreads();
  auto Gs = [&](int i, int j) { return (i <= j) ? G[gidx(i, j, K)] : G[gidx(j, i, K)]; };  // prescaled Gram
  auto gy = [&](int i) { return G[nG + i]; };
  auto us = [&](int i) { return G[nG + K + i]; };
  const double sy = G[nG + 2 * K + 1];

  // ---- old-state ridge OLS on XX = O / scale (no intercept): A_ij = Gs_ij / (s_i s_j scale^2)
  if (threadIdx.x == 0) {
    plan->anyflags = anyfl;
    plan->scale_fit = (anyfl & BSR_F_NAN) ? NAN : scale;
  }
  if (!anyfl) {
    if (threadIdx.x < RF_LDM * RF_LDM) {
      const int i = threadIdx.x / RF_LDM, j = threadIdx.x % RF_LDM;
      double a = (i == j) ? 1.0 : 0.0;
      if (i < K && j < K) {
        const double ti = 1.0 / (sc[i] * scale), tj = 1.0 / (sc[j] * scale);
        a = ti * (tj * Gs(i, j)) + ((i == j) ? 1e-6 : 0.0);
      }
      shA[threadIdx.x] = a;
    }
    __syncthreads();
    rf_gauss_jordan(shA, shI, K, &shpiv);
    if ((int)threadIdx.x < K) {
      double t = 0.0;
      for (int j = 0; j < K; ++j) t += shI[threadIdx.x * RF_LDM + j] * (gy(j) / (sc[j] * scale));
      plan->beta_fit[threadIdx.x] = t;
      plan->coef_fit[threadIdx.x] = t / scale;  // weight of the raw column in the fitted values
    }
    __syncthreads();
    // ---- intercept ridge OLS on [1 | O] / scale_i, scale_i = max(1, scale)
    const double scale_i = fmax(1.0, scale);
    if (threadIdx.x < RF_LDM * RF_LDM) {
      const int i = threadIdx.x / RF_LDM, j = threadIdx.x % RF_LDM;
      const int M = K + 1;
      double a = (i == j) ? 1.0 : 0.0;
      if (i < M && j < M) {
        double gij;
        if (i == 0 && j == 0) gij = (double)N / (scale_i * scale_i);
        else if (i == 0 || j == 0) {
          const int c = (i == 0 ? j : i) - 1;
          gij = us(c) / (sc[c] * scale_i) / scale_i;
        } else {
          gij = Gs(i - 1, j - 1) / (sc[i - 1] * scale_i) / (sc[j - 1] * scale_i);
        }
        a = gij + ((i == j) ? 1e-6 : 0.0);
      }
      shA[threadIdx.x] = a;
    }
    __syncthreads();
    rf_gauss_jordan(shA, shI, K + 1, &shpiv);
    if ((int)threadIdx.x <= K) {
      double t = 0.0;
      for (int j = 0; j <= K; ++j) {
        const double rhs = (j == 0) ? sy / scale_i : gy(j - 1) / (sc[j - 1] * scale_i);
        t += shI[threadIdx.x * RF_LDM + j] * rhs;
      }
      plan->beta_icpt[threadIdx.x] = t / scale_i;  // reference's Beta / scale: weight of the raw column
    }
  } else if (threadIdx.x <= (unsigned)K) {
    plan->beta_fit[threadIdx.x % RF_MAXK] = NAN;
    plan->coef_fit[threadIdx.x % RF_MAXK] = NAN;
    plan->beta_icpt[threadIdx.x] = NAN;
  }
  __syncthreads();
  // ---- per-k candidate prescale / sibling census (thread k), Cholesky of the whole prescaled Gram (thread 0)
  if ((int)threadIdx.x < K) {
    const int k = threadIdx.x;
    double m_other = 0.0;
    uint32_t fl = 0;
    for (int j = 0; j < K; ++j) {
      if (j == k) continue;
      m_other = fmax(m_other, in->colmax[j]);
      fl |= in->colflags[j] & (BSR_F_INF | BSR_F_NAN);
    }
    if (fl & BSR_F_INF) m_other = INFINITY;
    plan->s_k[k] = rf_pow2_prescale(m_other);
    plan->m_other[k] = m_other;
    plan->flags_k[k] = fl;
    plan->dcol[k] = sc[k];
  }
  if (threadIdx.x == 0) {
    for (int a = 0; a < K; ++a)
      for (int b = 0; b < K; ++b) sub[a * RF_NQ + b] = Gs(a, b);
    const double ratio = rf_chol_inv(sub, K, Rk, Tk);
    plan->fallback = (ratio > 1e-10) ? 0 : 1;
    plan->pad2 = 0;
    for (int i = 0; i < RF_NQ * RF_NQ; ++i) {
      plan->R1[i] = Rk[i];
      plan->T1[i] = Tk[i];
    }
  }
}

// r3 ---------------------------------------------------------------------------------------------------------------
// per block: words = K(K+1)/2 (Gram of Q1) + K (Q1^T y); plus 2 words for the two direct residuals
#define RF_AW 48
template <typename T, int K>
__global__ __launch_bounds__(RF_THREADS) void k_rf_apply1(const T* __restrict__ cols, T* __restrict__ Q,
                                                          const T* __restrict__ y, int64_t ld, int64_t N,
                                                          int rows_per_block, const RefreshPlan* __restrict__ plan,
                                                          double* __restrict__ part) {
  constexpr int nGq = K * (K + 1) / 2;
  constexpr int nW = nGq + K;
  __shared__ double sh[4 * nW];
  __shared__ double outv[nW];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(N, r0 + rows_per_block);
  double* o = part + (size_t)blockIdx.x * (RF_AW + 8);
  const bool fits = plan->anyflags == 0;
  // direct residuals of the two fits
  {
    double acc[2] = {0.0, 0.0};
    if (fits) {
      double cf[K], ci[K];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        cf[j] = plan->coef_fit[j];
        ci[j] = plan->beta_icpt[j + 1];
      }
      const double c0 = plan->beta_icpt[0];
      for (int64_t n = r0 + threadIdx.x; n < r1; n += RF_THREADS) {
        double f0 = 0.0, f1 = c0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const double v = (double)cols[(int64_t)j * ld + n];
          f0 = fma(cf[j], v, f0);
          f1 = fma(ci[j], v, f1);
        }
        const double yv = (double)y[n];
        acc[0] = fma(yv - f0, yv - f0, acc[0]);
        acc[1] = fma(yv - f1, yv - f1, acc[1]);
      }
    }
    rf_block_sum<2>(acc, sh, outv);
    if (threadIdx.x < 2) o[RF_AW + threadIdx.x] = outv[threadIdx.x];
    __syncthreads();
  }
  if (plan->fallback) return;
  double t1[nGq], dc[K];  // upper triangle of T1 and the column prescales (uniform: scalar registers)
  {
    int w = 0;
#pragma unroll
    for (int a_ = 0; a_ < K; ++a_)
#pragma unroll
      for (int b_ = a_; b_ < K; ++b_, ++w) t1[w] = plan->T1[a_ * RF_NQ + b_];
#pragma unroll
    for (int j = 0; j < K; ++j) dc[j] = plan->dcol[j];
  }
  double acc[nW];
#pragma unroll
  for (int q = 0; q < nW; ++q) acc[q] = 0.0;
  for (int64_t n = r0 + threadIdx.x; n < r1; n += RF_THREADS) {
    double v[K], q1[K];
#pragma unroll
    for (int a_ = 0; a_ < K; ++a_) v[a_] = (dc[a_] != 0.0) ? dc[a_] * (double)cols[(int64_t)a_ * ld + n] : 0.0;
#pragma unroll
    for (int b_ = 0; b_ < K; ++b_) {
      double t = 0.0;
#pragma unroll
      for (int a_ = 0; a_ <= b_; ++a_) t = fma(v[a_], t1[a_ * K - (a_ * (a_ - 1)) / 2 + (b_ - a_)], t);
      q1[b_] = t;
      Q[(int64_t)b_ * ld + n] = (T)t;
    }
    const double yv = (double)y[n];
    int w = 0;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int j = i; j < K; ++j, ++w) acc[w] = fma(q1[i], q1[j], acc[w]);
#pragma unroll
    for (int i = 0; i < K; ++i) acc[nGq + i] = fma(q1[i], yv, acc[nGq + i]);
  }
  rf_block_sum<nW>(acc, sh, outv);
  if (threadIdx.x < nW) o[threadIdx.x] = outv[threadIdx.x];
}

// r4 ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RF_THREADS) void k_rf_plan2(int K, int64_t N, int n_blocks, const double* __restrict__ part,
                                                         RefreshPlan* __restrict__ plan, ChainB* __restrict__ cb,
                                                         ChainFitOut* __restrict__ fit_noicpt,
                                                         ChainFitOut* __restrict__ fit_icpt,
                                                         const RefreshIn* __restrict__ in) {
  __shared__ double W[RF_AW + 8];
  __shared__ double sub[RF_NQ * RF_NQ], Rk[RF_NQ * RF_NQ], Tk[RF_NQ * RF_NQ];
  const int stride = RF_AW + 8;
  {
    __shared__ double red[4][RF_AW + 8];
    const int slice = threadIdx.x >> 6;
    for (int w = threadIdx.x & 63; w < stride; w += 64) {
      double t = 0.0;
#pragma unroll 4
      for (int b = slice; b < n_blocks; b += 4) t += part[(size_t)b * stride + w];
      red[slice][w] = t;
    }
    __syncthreads();
    for (int w = threadIdx.x; w < stride; w += RF_THREADS) W[w] = ((red[0][w] + red[1][w]) + red[2][w]) + red[3][w];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool ok = plan->anyflags == 0;
    fit_noicpt->sse = ok ? W[RF_AW] : NAN;
    fit_noicpt->scale = plan->scale_fit;
    fit_noicpt->anyflags = plan->anyflags;
    fit_icpt->sse = ok ? W[RF_AW + 1] : NAN;
    fit_icpt->scale = fmax(1.0, plan->scale_fit);
    fit_icpt->anyflags = plan->anyflags;
    for (int j = 0; j < K; ++j) {
      fit_noicpt->beta[j] = plan->beta_fit[j];
      fit_noicpt->maxabs[j] = in->colmax[j];
      fit_noicpt->colflags[j] = in->colflags[j];
    }
    for (int j = 0; j <= K; ++j) fit_icpt->beta_unscaled[j] = plan->beta_icpt[j];
    for (int k = 0; k < RF_MAXK; ++k) {
      cb->s_k[k] = (k < K) ? plan->s_k[k] : 1.0;
      cb->m_other[k] = (k < K) ? plan->m_other[k] : 0.0;
      cb->flags_k[k] = (k < K) ? plan->flags_k[k] : 0u;
      cb->d[k] = (k < K) ? plan->dcol[k] : 0.0;
      cb->qy[k] = 0.0;
    }
    for (int i = 0; i < RF_NQ * RF_NQ; ++i) cb->R[i] = 0.0;
    if (!plan->fallback) {
      const int nGq = K * (K + 1) / 2;
      for (int a = 0; a < K; ++a)
        for (int b = a; b < K; ++b) {
          const double v = W[gidx(a, b, K)];
          sub[a * RF_NQ + b] = v;
          sub[b * RF_NQ + a] = v;
        }
      const double ratio = rf_chol_inv(sub, K, Rk, Tk);
      if (!(ratio > 1e-6)) plan->fallback = 1;  // Q1 was far from orthonormal: the first factor was too inaccurate
      for (int i = 0; i < RF_NQ * RF_NQ; ++i) plan->T2[i] = Tk[i];
      // R = R2 R1 ; q^T y = T2^T (Q1^T y)
      for (int a = 0; a < K; ++a)
        for (int b = a; b < K; ++b) {
          double t = 0.0;
          for (int m = a; m <= b; ++m) t += Rk[a * RF_NQ + m] * plan->R1[m * RF_NQ + b];
          cb->R[a * RF_NQ + b] = t;
        }
      for (int b = 0; b < K; ++b) {
        double t = 0.0;
        for (int a = 0; a <= b; ++a) t += Tk[a * RF_NQ + b] * W[nGq + a];
        cb->qy[b] = t;
        plan->qy[b] = t;
      }
    }
  }
}

// r5 ---------------------------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(RF_THREADS) void k_rf_apply2(T* __restrict__ Q, const T* __restrict__ y, int64_t ld,
                                                          int64_t N, int rows_per_block,
                                                          const RefreshPlan* __restrict__ plan,
                                                          double* __restrict__ part) {
  constexpr int nGq = K * (K + 1) / 2;
  __shared__ double sh[4];
  __shared__ double outv[1];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(N, r0 + rows_per_block);
  double acc[1] = {0.0};
  if (plan->fallback) {  // the Gram-Schmidt kernel rebuilds everything, including |y_perp|^2
    if (threadIdx.x == 0) part[blockIdx.x] = 0.0;
    return;
  }
  double t2[nGq], qy[K];
  int w = 0;
#pragma unroll
  for (int a_ = 0; a_ < K; ++a_)
#pragma unroll
    for (int b_ = a_; b_ < K; ++b_, ++w) t2[w] = plan->T2[a_ * RF_NQ + b_];
#pragma unroll
  for (int b_ = 0; b_ < K; ++b_) qy[b_] = plan->qy[b_];
  for (int64_t n = r0 + threadIdx.x; n < r1; n += RF_THREADS) {
    double r = (double)y[n];
    double q1[K];
#pragma unroll
    for (int a_ = 0; a_ < K; ++a_) q1[a_] = (double)Q[(int64_t)a_ * ld + n];
#pragma unroll
    for (int b_ = 0; b_ < K; ++b_) {
      double t = 0.0;
#pragma unroll
      for (int a_ = 0; a_ <= b_; ++a_) t = fma(q1[a_], t2[a_ * K - (a_ * (a_ - 1)) / 2 + (b_ - a_)], t);
      const T qs = (T)t;
      Q[(int64_t)b_ * ld + n] = qs;
      r = fma(-qy[b_], (double)qs, r);
    }
    acc[0] = fma(r, r, acc[0]);
  }
  rf_block_sum<1>(acc, sh, outv);
  if (threadIdx.x == 0) part[blockIdx.x] = outv[0];
}

// r6 ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rf_final(int n_blocks, const double* __restrict__ part, ChainB* __restrict__ cb) {
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll 8
    for (int b = 0; b < n_blocks; ++b) t += part[b];
    cb->yperp2 = t;
  }
}

template <typename T, int K>
static void refresh_fast_k(hipStream_t st, const T* cols, T* Q, const T* y, int64_t ld, int64_t N,
                           const RefreshIn* d_in, RefreshPlan* d_plan, double* d_part, ChainB* cb,
                           ChainFitOut* fit_noicpt, ChainFitOut* fit_icpt) {
  const int rows = BSR_RF_ROWS;
  const int nb = (int)((N + rows - 1) / rows);
  hipLaunchKernelGGL((k_rf_gram<T, K>), dim3(nb), dim3(RF_THREADS), 0, st, cols, y, ld, N, rows, d_in, d_part);
  hipLaunchKernelGGL(k_rf_plan, dim3(1), dim3(RF_THREADS), 0, st, d_in, K, N, nb, d_part, d_plan);
  double* part2 = d_part + (size_t)nb * RF_GW;
  hipLaunchKernelGGL((k_rf_apply1<T, K>), dim3(nb), dim3(RF_THREADS), 0, st, cols, Q, y, ld, N, rows, d_plan, part2);
  hipLaunchKernelGGL(k_rf_plan2, dim3(1), dim3(RF_THREADS), 0, st, K, N, nb, part2, d_plan, cb, fit_noicpt, fit_icpt,
                     d_in);
  double* part3 = part2 + (size_t)nb * (RF_AW + 8);
  hipLaunchKernelGGL((k_rf_apply2<T, K>), dim3(nb), dim3(RF_THREADS), 0, st, Q, y, ld, N, rows, d_plan, part3);
  hipLaunchKernelGGL(k_rf_final, dim3(1), dim3(64), 0, st, nb, part3, cb);
}

template <typename T>
void launch_refresh_fast(hipStream_t st, const T* cols, T* Q, const T* y, int64_t ld, int64_t N, int K,
                         const RefreshIn* d_in, RefreshPlan* d_plan, double* d_part, ChainB* cb,
                         ChainFitOut* fit_noicpt, ChainFitOut* fit_icpt) {
#define RF_CASE(KK) \
  case KK: refresh_fast_k<T, KK>(st, cols, Q, y, ld, N, d_in, d_plan, d_part, cb, fit_noicpt, fit_icpt); break;
  switch (K) {
    RF_CASE(1) RF_CASE(2) RF_CASE(3) RF_CASE(4) RF_CASE(5) RF_CASE(6) RF_CASE(7) RF_CASE(8)
  }
#undef RF_CASE
}

size_t refresh_part_doubles(int64_t N) {
  const size_t nb = (size_t)((N + BSR_RF_ROWS - 1) / BSR_RF_ROWS);
  return nb * (RF_GW + RF_AW + 8 + 1);
}

template void launch_refresh_fast<double>(hipStream_t, const double*, double*, const double*, int64_t, int64_t, int,
                                          const RefreshIn*, RefreshPlan*, double*, ChainB*, ChainFitOut*, ChainFitOut*);
template void launch_refresh_fast<float>(hipStream_t, const float*, float*, const float*, int64_t, int64_t, int,
                                         const RefreshIn*, RefreshPlan*, double*, ChainB*, ChainFitOut*, ChainFitOut*);
