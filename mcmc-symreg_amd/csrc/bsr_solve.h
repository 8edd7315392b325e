// The per-proposal small algebra shared by k_solve / k_finalize / the fused finalise step (bsr_kernels.hip) and the row
// pass that solves a proposal itself when its last partial record has landed (bsr_tile_asm.hip): rank gate, ridge OLS,
// SSE and log-likelihood of codes/funcs.py:1147-1174, 1226 from the O(N) sums of the row pass.
#pragma once
#include "bsr_device.h"

// ---------------------------------------------------------------------------------------------------------------
// Per-proposal small algebra.  One wave per proposal.
//
// The chain keeps ONE orthonormal basis of its K current columns, O_j d_j = sum_i Q_i R_ij (d_j: power-of-two column
// prescales).  A proposal replaces tree k by the candidate z.  With s the candidate's prescale, c = Q^T (s z),
// w = s z - Q c (orthogonal to Q, |w| = rho):
//   s * new_outputs = [Q, w/rho] S,   S = [[R_{-k} (s/d), c], [0, rho]]      ((K+1) x K: R without column k)
// (columns: siblings ascending, then the candidate).  [Q, w/rho] has orthonormal columns, so the singular values of S
// are those of s*new_outputs (rank gate, codes/funcs.py:1226), and with XX = new_outputs/scale = [Q, w/rho] (tau S),
// tau = 1/(s*scale), h = [Q^T y, w.y/rho], S = U Sigma V^T (one-sided Jacobi on the K columns of length K+1):
//   Beta = V (tau Sigma)/(tau^2 Sigma^2 + 1e-6) U^T h                  (ridge OLS, codes/funcs.py:1151-1155)
//   SSE  = (|y_perp|^2 - (w.y/rho)^2) + | h - U diag(d_m/(d_m + 1e-6)) U^T h |^2,  d_m = tau^2 sigma_m^2
// (codes/funcs.py:1162): the first term is what lies outside the (K+1)-dimensional frame, the second the misfit
// inside it -- the ridge shrinkage plus the one frame direction the new columns do not span (the old column k's own
// contribution), measured as a residual vector, never as a difference of squares.  The only O(N) inputs are c,
// |s z|^2, s z.y from pass 1 -- or rho^2, w.y from the direct residual pass when rho^2 = |s z|^2 - |c|^2 would
// cancel (candidate nearly inside the span of the current columns).
#define BSR_SOLVE_CK_WORDS 104   // doubles of a k_solve wave's LDS copy of its chain's block (ChainB: 101)

struct SolveIn {
  const ChainB* ck;
  const double* c;   // LDS: projections of s*z on the basis (K values)
  double rho2;       // |w|^2
  double wy;         // w . y
  double zz;         // |s z|^2
  double tau, s, sigma, scale, maxabs;
  int K, k;
  int64_t N;
  uint32_t flags;
  double rank_floor;  // lower bound of the relative rank tolerance (0 in f64; a few eps_f32 when columns are f32)
  int exact;          // 1: singular values by Jacobi whatever the margin (BSR_SOLVE_EXACT=1; the standalone ylogLike)
  MhRes* mh;          // device-side copy of (loglik, rank) for the MH scan (k_events)
#ifdef BSR_SOLVE_STAMPS
  unsigned long long st[4];   // clock at the wave's start, behind the records' loads, behind their reduction, at solve_any
#endif
};
// -DBSR_SOLVE_STAMPS (a measuring build, never shipped: BSR_EXTRA_FLAGS of build.sh): a proposal settled by the fast tier
// returns, in beta[0..7], the shader-clock cycles its wave spent loading the partial records, reducing them, between
// that and the solve, in the solve's steps 1, 2 and 3, behind them, and in all (tools/probes/solve_stamps.py).
#ifdef BSR_SOLVE_STAMPS
#define BSR_SOLVE_STAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define BSR_SOLVE_STAMP(v)
#endif
template <int K>
__device__ __forceinline__ void solve_regs(const SolveIn& in, int lane, bsr_score* out);
template <int K>
__device__ __forceinline__ void solve_cols(const SolveIn& in, int lane, bsr_score* out);

// 1/x and 1/sqrt(x) for the Jacobi rotations: the hardware estimate and two Newton steps (a few roundings, not correctly
// rounded -- a rotation only has to be orthogonal to rounding, which cs^2 + sn^2 = 1 +- a few ulp is; the singular values
// themselves are measured with IEEE sqrt afterwards).  The IEEE division and square root are ~14 and ~20 dependent
// instructions each, five of them per rotation: two thirds of what a rotation costs a wave that runs alone.
__device__ __forceinline__ double rot_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double rot_rsq(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = fma(fma(-0.5 * x * r, r, 0.5), r, r);
  r = fma(fma(-0.5 * x * r, r, 0.5), r, r);
  return r;
}
// (cs, sn) of the rotation that annihilates gamma between columns of squared norms alpha and beta.
// The estimates' Newton steps turn the IEEE limits into NaN (rsq(inf) = 0, then -0.5 inf 0; rcp of a denormal is inf, then
// inf - inf): where zeta^2 overflows, or 2 gamma is too small to invert, the IEEE forms gave tt = 0 -- the identity
// rotation, which is also the right answer (the off-diagonal is negligible next to the diagonal's gap) -- so anything
// that is not finite falls back to exactly that.  Reachable only with column norms ~1e270 apart; a NaN here would
// spread into W and every score of the proposal.
__device__ __forceinline__ void rot_coeffs(double alpha, double beta, double gamma, double* cs, double* sn) {
  const double zeta = (beta - alpha) * rot_rcp(2.0 * gamma);
  const double s1 = fma(zeta, zeta, 1.0);
  const double tt = copysign(1.0, zeta) * rot_rcp(fabs(zeta) + s1 * rot_rsq(s1));
  const double c0 = rot_rsq(fma(tt, tt, 1.0));
  const double s0 = c0 * tt;
  const bool ok = isfinite(c0) && isfinite(s0);
  *cs = ok ? c0 : 1.0;
  *sn = ok ? s0 : 0.0;
}

// s / d for d a power of two (the columns' prescales are: bsr_refresh.hip): exact like the IEEE quotient, by the exponent --
// two instructions instead of the division's thirty-odd, once per factor entry
__device__ __forceinline__ double pow2_quot(double s, double d) { return ldexp(s, 1 - __builtin_amdgcn_frexp_exp(d)); }

// entry (i, m) of the (K+1) x K factor S, m < K-1 a sibling column, m == K-1 the candidate.  Branch-free on purpose: written
// with the loads inside the conditions the compiler made every entry a branch with two dependent LDS reads waited for one
// by one (-DBSR_SOLVE_STAMPS: 8 000 of a K = 8 wave's 21 000 cycles went into filling the factor); this way the reads
// of all entries go out together and a select picks the value -- the same value.
__device__ __forceinline__ double factor_entry(const SolveIn& in, int K, int i, int m, double rho) {
  if (m == K - 1) return (i < K) ? in.c[i < K ? i : 0] : rho;
  const int j = (m < in.k) ? m : m + 1;            // sibling tree (j <= K - 1: inside the block whatever k is)
  const double dj = in.ck->d[j];
  const double r = in.ck->R[(i < K ? i : 0) * BSR_NQ_MAX + j];
  const double sc = pow2_quot(in.s, dj);           // (dj == 0: a number nobody reads)
  const bool live = (i <= j) & (i < K) & (dj != 0.0);
  return live ? r * sc : 0.0;                      // (a column that holds inf / NaN enters as zeros: d_j = 0)
}

__device__ __forceinline__ void store_score(const SolveIn& in, int K, bsr_score* out, double ll, double sse, double smin,
                                            double smax, int rank, const double* bt) {
  out->loglik = ll;
  out->sse = sse;
  out->scale = in.scale;
  out->maxabs = in.maxabs;
  out->smin = smin / in.s;
  out->smax = smax / in.s;
  out->rank = rank;
  out->flags = in.flags | ((rank < K) ? BSR_F_RANKDEF : 0u);
  in.mh->loglik = ll;
  in.mh->rank = rank;
}

// K <= 4: the factor, its one-sided Jacobi SVD (W = S V ends with mutually orthogonal columns W[:,m] = sigma_m u_m)
// and the ridge formulas held entirely in registers (K is a template parameter: every index is static).  All lanes
// compute the same values, so there is no cross-lane traffic at all.
template <int K>
__device__ __forceinline__ void solve_regs(const SolveIn& in, int lane, bsr_score* out) {
  constexpr int M = K + 1;
  const int k = in.k;
  const ChainB* ck = in.ck;
  // a residual at the rounding level of s z (a candidate that reproduces a vector of the span, e.g. the old column
  // itself) carries no direction: w = 0.  The cut is half of numpy's rank tolerance relative to |s z|, so it moves
  // every singular value of the factor by less than half that tolerance.
  const double rel_w = 0.5 * (double)((in.N > (int64_t)K) ? in.N : (int64_t)K) * 2.220446049250313e-16;
  const double rho = (in.rho2 > fmax(1e-30, rel_w * rel_w) * in.zz) ? sqrt(in.rho2) : 0.0;
  double W[M][K], V[K][K];
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) W[i][j] = factor_entry(in, K, i, j, rho);
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int a = 0; a < K - 1; ++a) {
#pragma unroll
      for (int b = a + 1; b < K; ++b) {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          alpha = fma(W[i][a], W[i][a], alpha);
          beta = fma(W[i][b], W[i][b], beta);
          gamma = fma(W[i][a], W[i][b], gamma);
        }
        // the rotation is skipped when |gamma| <= 1e-17 sqrt(alpha beta) and the sweep's measure is max |gamma| /
        // sqrt(alpha beta): both compared in squares (two dependent square roots fewer per rotation)
        const double ab = alpha * beta, g2 = gamma * gamma;
        if (ab > 0.0 && g2 > 1e-34 * ab) {
          off = fmax(off, g2 * rot_rcp(ab));
          double cs, sn;
          rot_coeffs(alpha, beta, gamma, &cs, &sn);
#pragma unroll
          for (int i = 0; i < M; ++i) {
            const double wa = W[i][a], wb = W[i][b];
            W[i][a] = cs * wa - sn * wb;
            W[i][b] = sn * wa + cs * wb;
          }
#pragma unroll
          for (int i = 0; i < K; ++i) {
            const double va = V[i][a], vb = V[i][b];
            V[i][a] = cs * va - sn * vb;
            V[i][b] = sn * va + cs * vb;
          }
        }
      }
    }
    // (`off` is what the sweep FOUND in front of its rotations, so the last sweep only looks.  Stopping at 1e-18 --
    // quadratic convergence: a sweep that found 1e-9 relative leaves 1e-18 -- was tried in round 4: no measurable
    // difference, 8.85 against 8.82 us at K = 3, 22.6 against 23.1 at K = 8.  The kernel's 8.8 us are 4.6 us of launch,
    // partial-record loads and lane reductions and 4.2 us of solve, most of it the divisions, square roots and the
    // logarithm behind the sweeps.)
    if (off <= 1e-30) break;
  }
  double h[M];
#pragma unroll
  for (int i = 0; i < M; ++i) h[i] = (i < K) ? ck->qy[i < K ? i : 0] : ((rho > 0.0) ? in.wy / rho : 0.0);
  const double eps = 1e-6;
  double sv[K], coefj[K], r[M];
#pragma unroll
  for (int i = 0; i < M; ++i) r[i] = h[i];
  double smax = 0.0, smin = INFINITY;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    double n2 = 0.0, tj = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      n2 = fma(W[i][j], W[i][j], n2);
      tj = fma(W[i][j], h[i], tj);
    }
    sv[j] = sqrt(n2);
    const double aj = (sv[j] > 0.0) ? tj / sv[j] : 0.0;                // u_j . h
    const double dj = (in.tau * sv[j]) * (in.tau * sv[j]);
    coefj[j] = (in.tau * sv[j]) / (dj + eps) * aj;
    const double gj = (sv[j] > 0.0) ? dj / (dj + eps) * aj / sv[j] : 0.0;  // fitted share along u_j, per unit of W[:,j]
#pragma unroll
    for (int i = 0; i < M; ++i) r[i] = fma(-gj, W[i][j], r[i]);
    smax = fmax(smax, sv[j]);
    smin = fmin(smin, sv[j]);
  }
  double misfit = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i) misfit = fma(r[i], r[i], misfit);
  const double dimmax = (double)((in.N > (int64_t)K) ? in.N : (int64_t)K);
  const double tol = smax * fmax(dimmax * 2.220446049250313e-16, in.rank_floor);
  int rank = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) rank += (sv[j] > tol) ? 1 : 0;
  const double sse = fmax(0.0, ck->yperp2 - h[K] * h[K]) + misfit;
  const double sigma = in.sigma;
  const double ll = -sse / (2 * sigma * sigma) - 0.5 * (double)in.N * log(2 * M_PI * sigma * sigma);
  double bt[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    double bi = 0.0;
#pragma unroll
    for (int j = 0; j < K; ++j) bi = fma(V[i][j], coefj[j], bi);
    bt[i] = bi;
  }
  if (lane == 0) {
    store_score(in, K, out, ll, sse, smin, smax, rank, bt);
#pragma unroll
    for (int i = 0; i < BSR_MAX_K; ++i) out->beta[i] = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int tree = (i == K - 1) ? k : ((i < k) ? i : i + 1);
      out->beta[tree] = bt[i];
    }
  }
}

// K = 5..8: eight lanes per proposal, lane j = lane & 7 owns column j of the (K+1) x K factor W and of V in registers
// (the 8 lane groups of the wave hold identical copies).  One-sided Jacobi with the XOR tournament ordering: in round
// r = 1..7 column j pairs with column j ^ r, so the four rotations of a round run side by side and a sweep is 7
// dependent steps instead of 28; the only cross-lane traffic is the partner's column (2K+1 shuffles per round), every
// dot product is lane-local.  Both lanes of a pair evaluate the same expressions on the same operands, so they apply
// bit-identical rotation coefficients.
// v of lane (l ^ r), r in 1..7, inside groups of eight lanes -- with DPP moves (a couple of cycles) instead of
// __shfl_xor, which lowers to ds_bpermute (an LDS round trip per 32 bits; the K >= 5 solver issues 17 of them per
// rotation).  quad_perm covers r = 1, 2, 3; row_half_mirror is l -> 7 - l = l ^ 7; the rest are two steps.
__device__ __forceinline__ double xor8(double v, int r) {
  switch (r) {
    case 1: return dpp_f64<0xB1, 0xF>(v);
    case 2: return dpp_f64<0x4E, 0xF>(v);
    case 3: return dpp_f64<0x1B, 0xF>(v);
    case 4: return dpp_f64<0x1B, 0xF>(dpp_f64<0x141, 0xF>(v));
    case 5: return dpp_f64<0x4E, 0xF>(dpp_f64<0x141, 0xF>(v));
    case 6: return dpp_f64<0xB1, 0xF>(dpp_f64<0x141, 0xF>(v));
    default: return dpp_f64<0x141, 0xF>(v);
  }
}

template <int K>
__device__ __forceinline__ void solve_cols(const SolveIn& in, int lane, bsr_score* out) {
  constexpr int M = K + 1;
  const int j = lane & 7;
  const int k = in.k;
  const ChainB* ck = in.ck;
  // a residual at the rounding level of s z (a candidate that reproduces a vector of the span, e.g. the old column
  // itself) carries no direction: w = 0.  The cut is half of numpy's rank tolerance relative to |s z|, so it moves
  // every singular value of the factor by less than half that tolerance.
  const double rel_w = 0.5 * (double)((in.N > (int64_t)K) ? in.N : (int64_t)K) * 2.220446049250313e-16;
  const double rho = (in.rho2 > fmax(1e-30, rel_w * rel_w) * in.zz) ? sqrt(in.rho2) : 0.0;
  double W[M], V[K];
#pragma unroll
  for (int i = 0; i < M; ++i) W[i] = (j < K) ? factor_entry(in, K, i, (j < K) ? j : 0, rho) : 0.0;
#pragma unroll
  for (int i = 0; i < K; ++i) V[i] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      const int hb = (r >= 4) ? 4 : ((r >= 2) ? 2 : 1);
      const bool low = (j & hb) == 0;  // j < (j ^ r)
      double Wp[M], Vp[K];
      double mine = 0.0, theirs = 0.0, gamma = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) Wp[i] = xor8(W[i], r);
#pragma unroll
      for (int i = 0; i < K; ++i) Vp[i] = xor8(V[i], r);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        mine = fma(W[i], W[i], mine);
        theirs = fma(Wp[i], Wp[i], theirs);
        gamma = fma(W[i], Wp[i], gamma);
      }
      const double alpha = low ? mine : theirs, beta = low ? theirs : mine;
      const double ab = alpha * beta, g2 = gamma * gamma;   // squared criteria: see solve_regs
      if (ab > 0.0 && g2 > 1e-34 * ab) {
        off = fmax(off, g2 * rot_rcp(ab));
        double cs, sn;
        rot_coeffs(alpha, beta, gamma, &cs, &sn);
        const double sp = low ? -sn : sn;  // low column: cs*W - sn*Wp ; high column: sn*Wp + cs*W
#pragma unroll
        for (int i = 0; i < M; ++i) W[i] = cs * W[i] + sp * Wp[i];
#pragma unroll
        for (int i = 0; i < K; ++i) V[i] = cs * V[i] + sp * Vp[i];
      }
    }
    off = fmax(off, xor8(off, 1));
    off = fmax(off, xor8(off, 2));
    off = fmax(off, xor8(off, 4));
    if (off <= 1e-30) break;  // wave-uniform: the lane groups are copies of each other
  }
  double h[M];
#pragma unroll
  for (int i = 0; i < M; ++i) h[i] = (i < K) ? ck->qy[i < K ? i : 0] : ((rho > 0.0) ? in.wy / rho : 0.0);
  double n2 = 0.0, tj = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    n2 = fma(W[i], W[i], n2);
    tj = fma(W[i], h[i], tj);
  }
  const double eps = 1e-6;
  const double sv = sqrt(n2);                                   // sigma_j
  const double aj = (sv > 0.0 && j < K) ? tj / sv : 0.0;        // u_j . h
  const double dj = (in.tau * sv) * (in.tau * sv);
  const double coef = (j < K) ? (in.tau * sv) / (dj + eps) * aj : 0.0;
  const double gj = (sv > 0.0 && j < K) ? dj / (dj + eps) * aj / sv : 0.0;
  // residual inside the frame: r = h - sum_j u_j (d_j/(d_j+eps)) (u_j . h), summed over the 8 column lanes
  double misfit = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    double t = gj * W[i];
    t += xor8(t, 1);
    t += xor8(t, 2);
    t += xor8(t, 4);
    const double ri = h[i] - t;
    misfit = fma(ri, ri, misfit);
  }
  double smax = 0.0, smin = INFINITY;
#pragma unroll
  for (int a = 0; a < K; ++a) smax = fmax(smax, __shfl(sv, a, 8));
  const double dimmax = (double)((in.N > (int64_t)K) ? in.N : (int64_t)K);
  const double tol = smax * fmax(dimmax * 2.220446049250313e-16, in.rank_floor);  // numpy matrix_rank default
  int rank = 0;
#pragma unroll
  for (int a = 0; a < K; ++a) {
    const double sva = __shfl(sv, a, 8);
    smin = fmin(smin, sva);
    rank += (sva > tol) ? 1 : 0;
  }
  const double sse = fmax(0.0, ck->yperp2 - h[K] * h[K]) + misfit;
  const double sigma = in.sigma;
  const double ll = -sse / (2 * sigma * sigma) - 0.5 * (double)in.N * log(2 * M_PI * sigma * sigma);
  double bt[K];                                                 // Beta'_i = sum_j V[i][j] coef_j
#pragma unroll
  for (int i = 0; i < K; ++i) {
    double b = V[i] * coef;
    b += xor8(b, 1);
    b += xor8(b, 2);
    b += xor8(b, 4);
    bt[i] = b;
  }
  if (lane == 0) {
    store_score(in, K, out, ll, sse, smin, smax, rank, bt);
#pragma unroll
    for (int i = 0; i < BSR_MAX_K; ++i) out->beta[i] = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int tree = (i == K - 1) ? k : ((i < k) ? i : i + 1);
      out->beta[tree] = bt[i];
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The fast tier (round 6).  Almost every proposal is far from the rank gate's threshold, on one side or the other, and
// the (K+1) x K factor S is nearly triangular already: its first k columns ARE (R's own), columns k..K-2 carry one
// subdiagonal entry (R with column k taken out is upper Hessenberg) and only the candidate's column is dense.  So:
//   1. K - k Givens rotations of adjacent rows make S = G [T; 0], T upper triangular K x K; g = G^T h.
//   2. The gate from T alone, by bounds: sigma_min <= min |t_jj| and sigma_max >= the largest column norm -- a quarter
//      of the tolerance below => rank < K for certain; sigma_min >= 1 / |T^-1|_F and sigma_max <= |T|_F (T^-1 by back
//      substitution) -- four tolerances above => rank = K for certain.  In between (a band a few hundred wide around
//      N eps, where also numpy's own SVD of the N x K matrix is within rounding of its threshold) and for anything
//      that is not finite: the one-sided Jacobi SVD below, as for every proposal until round 5.
//   3. rank = K: the ridge fit min |g1 - tau T b|^2 + 1e-6 |b|^2 (codes/funcs.py:1151-1155) by K Householder reflections
//      of [tau T; 1e-3 I] (backward stable at any condition number the gate lets through, no normal equations; until
//      the middle of round 6 K (K + 1) / 2 Givens rotations: twice the instructions at K = 8), b by back substitution, the misfit as the residual VECTOR g1 - tau T b plus the frame
//      direction the columns do not span (g_K) -- the quantities of the Jacobi path, computed from a QR instead of an SVD.
//   4. rank < K: the reference returns before ylogLike (codes/funcs.py:1226-1228): there is no log-likelihood to
//      report -- loglik, sse and beta are NaN, rank is a bound (the diagonal entries above the tolerance, at most
//      K - 1), BSR_F_SV_BOUNDS says so.  (The standalone ylogLike entry point, which has no gate, asks for the exact
//      tier: bit 1 of PropDesc::self_dup.)
// smin / smax of a proposal settled here are the bounds used (within sqrt(K) of the singular values), flagged
// BSR_F_SV_BOUNDS.  All lanes compute the same values; every index is static.
// A dependent chain of ~40 (K = 3) to ~400 (K = 8) instructions instead of five or six Jacobi sweeps of 3 x ~90 / 7 x ~150.
template <int K>
__device__ __forceinline__ bool solve_fast(const SolveIn& in, int lane, bsr_score* out) {
  constexpr int M = K + 1;
  const int k = in.k;
  const ChainB* ck = in.ck;
  const double rel_w = 0.5 * (double)((in.N > (int64_t)K) ? in.N : (int64_t)K) * 2.220446049250313e-16;
  const double rho = (in.rho2 > fmax(1e-30, rel_w * rel_w) * in.zz) ? sqrt(in.rho2) : 0.0;
  double W[M][K], g[M];
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) W[i][j] = factor_entry(in, K, i, j, rho);
#pragma unroll
  for (int i = 0; i < M; ++i) g[i] = (i < K) ? ck->qy[i < K ? i : 0] : ((rho > 0.0) ? in.wy / rho : 0.0);
  const double hK = g[K];
  double colmax2 = 0.0;   // the largest squared column norm: sigma_max^2 is at least that
#pragma unroll
  for (int j = 0; j < K; ++j) {
    double n2 = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) n2 = fma(W[i][j], W[i][j], n2);
    colmax2 = fmax(colmax2, n2);
  }
  BSR_SOLVE_STAMP(t_a);
  // 1. rows (m, m + 1) for the Hessenberg columns m = k..K-2, then rows (K-1, K) for the candidate's column
#pragma unroll
  for (int m = 0; m < K; ++m) {
    if (m >= k || m == K - 1) {
      const double a = W[m][m], b = W[m + 1][m];
      if (b != 0.0) {
        const double n2 = fma(a, a, b * b);
        const double ri = rot_rsq(n2);
        const double cs = a * ri, sn = b * ri;
        W[m][m] = n2 * ri;
        W[m + 1][m] = 0.0;
#pragma unroll
        for (int j = m + 1; j < K; ++j) {
          const double t0 = W[m][j], t1 = W[m + 1][j];
          W[m][j] = fma(cs, t0, sn * t1);
          W[m + 1][j] = fma(cs, t1, -(sn * t0));
        }
        const double g0 = g[m], g1 = g[m + 1];
        g[m] = fma(cs, g0, sn * g1);
        g[m + 1] = fma(cs, g1, -(sn * g0));
      }
    }
  }
  BSR_SOLVE_STAMP(t_b);
  // 2. T^-1 (upper triangular) by back substitution, the bounds, the verdict
  double Y[K][K];
  double mind = INFINITY, frobT2 = 0.0, frobY2 = 0.0;
  {
    // (four partial sums: one accumulator made the K (K + 1) / 2 squares a single dependent chain, ~9.5 cycles a link for a
    // wave that runs alone)
    double f4[4] = {0.0, 0.0, 0.0, 0.0};
    int n = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      mind = fmin(mind, fabs(W[j][j]));
#pragma unroll
      for (int i = 0; i <= j; ++i) {
        f4[n & 3] = fma(W[i][j], W[i][j], f4[n & 3]);
        ++n;
      }
    }
    frobT2 = (f4[0] + f4[1]) + (f4[2] + f4[3]);
  }
  const double dimmax = (double)((in.N > (int64_t)K) ? in.N : (int64_t)K);
  const double tolrel = fmax(dimmax * 2.220446049250313e-16, in.rank_floor);
  // (bounds: the reciprocal square root's estimate and two Newton steps, a third of the IEEE square root's chain)
  const double smax_lb = colmax2 * rot_rsq(colmax2), smax_ub = frobT2 * rot_rsq(frobT2);
  if (!(isfinite(smax_ub) && smax_ub > 0.0)) return false;
  if (mind < 0.25 * tolrel * smax_lb) {   // 4. rank < K for certain
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) cnt += (fabs(W[j][j]) > tolrel * smax_ub) ? 1 : 0;
    if (cnt > K - 1) cnt = K - 1;
    if (lane == 0) {
      out->loglik = NAN;
      out->sse = NAN;
      out->scale = in.scale;
      out->maxabs = in.maxabs;
      out->smin = mind / in.s;
      out->smax = smax_ub / in.s;
      out->rank = cnt;
      out->flags = in.flags | BSR_F_RANKDEF | BSR_F_SV_BOUNDS;
      in.mh->loglik = NAN;
      in.mh->rank = cnt;
#pragma unroll
      for (int i = 0; i < BSR_MAX_K; ++i) out->beta[i] = (i < K) ? NAN : 0.0;
    }
    return true;
  }
#pragma unroll
  for (int j = 0; j < K; ++j) Y[j][j] = rot_rcp(W[j][j]);   // (the estimate and two Newton steps: ~1 ulp, a third of the IEEE division's chain)
#pragma unroll
  for (int j = 0; j < K; ++j) {
#pragma unroll
    for (int i = j - 1; i >= 0; --i) {
      double acc = 0.0;
#pragma unroll
      for (int l = i + 1; l <= j; ++l) acc = fma(W[i][l], Y[l][j], acc);
      Y[i][j] = -(acc * Y[i][i]);
    }
  }
  {
    double f4[4] = {0.0, 0.0, 0.0, 0.0};
    int n = 0;
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
      for (int i = 0; i <= j; ++i) {
        f4[n & 3] = fma(Y[i][j], Y[i][j], f4[n & 3]);
        ++n;
      }
    frobY2 = (f4[0] + f4[1]) + (f4[2] + f4[3]);
  }
  const double smin_lb = rot_rsq(frobY2);   // 1 / |T^-1|_F
  if (!(smin_lb > 4.0 * tolrel * smax_ub)) return false;   // the band around the tolerance (or NaN): the exact tier decides
  BSR_SOLVE_STAMP(t_c);
  // 3. ridge: [tau T; sqrt(1e-6) I | g1; 0] to triangular form, column by column, by Householder reflections.  Column
  //    j's entries to remove sit in the identity's rows 0..j (row j's own 1e-3 and what the earlier columns' reflections
  //    filled in), so one reflection over (T row j, identity rows 0..j) does what j + 1 rotations did until this round:
  //    one square root and one reciprocal per COLUMN instead of one square root per ENTRY -- ~500 instead of ~1 050
  //    instructions at K = 8 (every lane runs the whole chain: the instruction count is what the wave pays), as
  //    backward stable as the rotations, no normal equations.  v = x + sign(a) |x| e1 (no cancellation).
  const double tau = in.tau;
  double T2[K][K], g2[K], invd[K];
  double E[K][K], ge[K];   // the identity's rows as they fill in: E[e][l] for l > the column being worked on; their right-hand sides
#pragma unroll
  for (int i = 0; i < K; ++i) {
    g2[i] = g[i];
    ge[i] = 0.0;
#pragma unroll
    for (int j = 0; j < K; ++j) E[i][j] = (j == i) ? 1e-3 : 0.0;
#pragma unroll
    for (int j = i; j < K; ++j) T2[i][j] = tau * W[i][j];
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const double a = T2[j][j];
    double s2 = 0.0;
#pragma unroll
    for (int e = 0; e <= j; ++e) s2 = fma(E[e][j], E[e][j], s2);
    const double n2 = fma(a, a, s2);
    const double nrm = n2 * rot_rsq(n2);            // |x|
    const double v0 = a + copysign(nrm, a);         // v = (v0, E[0..j][j])
    const double coef = rot_rcp(nrm * fabs(v0));    // 2 / v.v = 1 / (|x| (|x| + |a|))
    const double alpha = -copysign(nrm, a);         // the column's new diagonal entry
    T2[j][j] = alpha;
    invd[j] = rot_rcp(alpha);
#pragma unroll
    for (int l = j + 1; l < K; ++l) {
      double w = v0 * T2[j][l];
#pragma unroll
      for (int e = 0; e <= j; ++e) w = fma(E[e][j], E[e][l], w);
      w *= coef;
      T2[j][l] = fma(-w, v0, T2[j][l]);
#pragma unroll
      for (int e = 0; e <= j; ++e) E[e][l] = fma(-w, E[e][j], E[e][l]);
    }
    double wg = v0 * g2[j];
#pragma unroll
    for (int e = 0; e <= j; ++e) wg = fma(E[e][j], ge[e], wg);
    wg *= coef;
    g2[j] = fma(-wg, v0, g2[j]);
#pragma unroll
    for (int e = 0; e <= j; ++e) ge[e] = fma(-wg, E[e][j], ge[e]);
  }
  BSR_SOLVE_STAMP(t_d);
  double bt[K];
#pragma unroll
  for (int i = K - 1; i >= 0; --i) {
    double acc = g2[i];
#pragma unroll
    for (int l = i + 1; l < K; ++l) acc = fma(-T2[i][l], bt[l], acc);
    bt[i] = acc * invd[i];
  }
  double misfit = g[K] * g[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    double r = g[i];
#pragma unroll
    for (int l = i; l < K; ++l) r = fma(-(tau * W[i][l]), bt[l], r);
    misfit = fma(r, r, misfit);
  }
  const double sse = fmax(0.0, ck->yperp2 - hK * hK) + misfit;
  const double sigma = in.sigma;
  const double ll = -sse / (2 * sigma * sigma) - 0.5 * (double)in.N * log(2 * M_PI * sigma * sigma);
  if (!isfinite(ll)) return false;
  if (lane == 0) {
    out->loglik = ll;
    out->sse = sse;
    out->scale = in.scale;
    out->maxabs = in.maxabs;
    out->smin = smin_lb / in.s;
    out->smax = smax_ub / in.s;
    out->rank = K;
    out->flags = in.flags | BSR_F_SV_BOUNDS;
    in.mh->loglik = ll;
    in.mh->rank = K;
#pragma unroll
    for (int i = 0; i < BSR_MAX_K; ++i) out->beta[i] = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int tree = (i == K - 1) ? k : ((i < k) ? i : i + 1);
      out->beta[tree] = bt[i];
    }
#ifdef BSR_SOLVE_STAMPS
    {
      const unsigned long long t_e = __builtin_readcyclecounter();
      out->beta[0] = (double)(in.st[1] - in.st[0]);
      out->beta[1] = (double)(in.st[2] - in.st[1]);
      out->beta[2] = (double)(t_a - in.st[2]);
      out->beta[3] = (double)(t_b - t_a);
      out->beta[4] = (double)(t_c - t_b);
      out->beta[5] = (double)(t_d - t_c);
      out->beta[6] = (double)(t_e - t_d);
      out->beta[7] = (double)(t_e - in.st[0]);
    }
#endif
  }
  return true;
}

__device__ __forceinline__ void solve_any(const SolveIn& in, int lane, bsr_score* out) {
  if (!in.exact) {   // (wave-uniform: every lane holds the same numbers)
    bool done;
    switch (in.K) {
      case 1: done = solve_fast<1>(in, lane, out); break;
      case 2: done = solve_fast<2>(in, lane, out); break;
      case 3: done = solve_fast<3>(in, lane, out); break;
      case 4: done = solve_fast<4>(in, lane, out); break;
      case 5: done = solve_fast<5>(in, lane, out); break;
      case 6: done = solve_fast<6>(in, lane, out); break;
      case 7: done = solve_fast<7>(in, lane, out); break;
      default: done = solve_fast<8>(in, lane, out); break;
    }
    if (done) return;
  }
  switch (in.K) {
    case 1: solve_regs<1>(in, lane, out); break;
    case 2: solve_regs<2>(in, lane, out); break;
    case 3: solve_regs<3>(in, lane, out); break;
    case 4: solve_regs<4>(in, lane, out); break;
    case 5: solve_cols<5>(in, lane, out); break;
    case 6: solve_cols<6>(in, lane, out); break;
    case 7: solve_cols<7>(in, lane, out); break;
    default: solve_cols<8>(in, lane, out); break;
  }
}


// One proposal, one wave: reduces its n_rb partial records of the row pass (fixed order: lane-strided, then the wave
// reduction), then rank gate / OLS / log-likelihood, or -- a candidate (nearly) inside the span of the current columns --
// hands it to the residual pass through the flagged list.  sh_c: BSR_NQ_MAX doubles of LDS of the wave's own.
// UNCACHED: the records sit in uncached memory and were written by other waves of THIS launch (read around the caches).
template <bool UNCACHED, bool STAGED = false>
__device__ __forceinline__ void solve_proposal(const PropDesc CONSTANT_AS* dsc, const ChainB* __restrict__ cks, int p, int n_rb,
                                               const double* __restrict__ part1, int64_t N, PropCoef* __restrict__ coef,
                                               bsr_score* __restrict__ outv, double rank_floor, int32_t* __restrict__ flagged,
                                               MhRes* __restrict__ mhv, int lane, double* sh_c, double* sh_ck = nullptr) {
  BSR_SOLVE_STAMP(t_s0);
  // The chain's block (R, Q^T y, prescales: 101 words) is read some forty times below, entry by entry at indices that
  // depend on k -- as loads from device memory each waited for on its own that was 5 of k_solve's 11 us at K = 8
  // (-DBSR_SOLVE_STAMPS).  STAGED (k_solve, sh_ck: BSR_SOLVE_CK_WORDS doubles of LDS of the wave's own): the wave fetches the block once, two words per lane, under the partial
  // records' own flight, and everything behind reads its copy in LDS.
  // (BSR_SOLVE_CK_WORDS doubles per wave: with the waves' projections 3.6 KB per workgroup -- what must fit the 4 KB of LDS
  // a tile or streaming workgroup of the NEXT batch leaves free on its CU.  With 6 KB -- three words per lane, the first
  // version -- k_solve waited for those workgroups to end wherever they take all they may: config 5's fp32 step went from
  // 69.5 to 77.6 us.)
  constexpr int CK_WORDS = (int)(sizeof(ChainB) / sizeof(double));
  static_assert(sizeof(ChainB) % sizeof(double) == 0 && CK_WORDS <= BSR_SOLVE_CK_WORDS && BSR_SOLVE_CK_WORDS <= 2 * BSR_WAVE,
                "two words per lane hold a ChainB");
  const ChainB* ck_dev = cks + dsc[p].ck;
  // (every field of the descriptor that is used below, asked for here: with a uniform p -- k_solve -- they are scalar
  // loads that travel together with the first one instead of one round trip each where they are used)
  const int d_mode = dsc[p].mode, d_K = dsc[p].K, d_k = dsc[p].k, d_nq = dsc[p].nq, d_dup = dsc[p].self_dup;
  const double d_s = dsc[p].s, d_sigma = dsc[p].sigma;
  double sum[BSR_NQ_MAX + 2];
#pragma unroll
  for (int i = 0; i < BSR_NQ_MAX + 2; ++i) sum[i] = 0.0;
  double amax = 0.0;
  uint32_t fl = 0;
  // One turn = two records per lane, both in flight together (the sums take them in the order of the plain loop).
  // Selects, not branches: behind a branch the compiler sinks the second record's loads and they wait their turn; x + 0.0
  // leaves every x as it is (the sums start at +0.0 and can never be -0.0).  The census of a partial record: inf in a
  // row <=> its max|z| is inf; NaN in a row <=> its |s z|^2 is NaN (the tile pass leaves word 11 zero and the census to
  // these two tests; the work-queue pass also sets the bits itself).
  auto take = [&](const double (&q)[BSR_P1_WORDS], bool valid) {
#pragma unroll
    for (int i = 0; i < BSR_NQ_MAX + 2; ++i) sum[i] += valid ? q[i] : 0.0;
    amax = fmax(amax, valid ? q[10] : 0.0);
    fl |= valid ? ((uint32_t)q[11] | ((q[10] == INFINITY) ? BSR_F_INF : 0u) | (isnan(q[8]) ? BSR_F_NAN : 0u)) : 0u;
  };
  auto fetch = [&](double (&q)[BSR_P1_WORDS], int rb) {
    const double* qp = part1 + ((size_t)p * n_rb + rb) * BSR_P1_WORDS;
#pragma unroll
    for (int i = 0; i < BSR_P1_WORDS; ++i) q[i] = UNCACHED ? __builtin_nontemporal_load(qp + i) : qp[i];
  };
  double ckw[2] = {0.0, 0.0};
  {
    // the first turn's loads go out in front of everything that needs the descriptor (their addresses come from the
    // kernel's arguments alone): the chain's block, which does, then travels with them instead of a round trip ahead
    const bool v1 = lane < n_rb, v2 = lane + BSR_WAVE < n_rb;
    double q[BSR_P1_WORDS], q2[BSR_P1_WORDS];
    fetch(q, v1 ? lane : 0);
    fetch(q2, v2 ? lane + BSR_WAVE : 0);
    if constexpr (STAGED) {
      if (d_mode != BSR_MODE_EVAL) {   // (an evaluate-only tape belongs to no chain: no block to fetch -- its index is -1)
        const double* src = reinterpret_cast<const double*>(ck_dev);
#pragma unroll
        for (int w = 0; w < 2; ++w)
          if (lane + w * BSR_WAVE < CK_WORDS) ckw[w] = src[lane + w * BSR_WAVE];
      }
    }
    take(q, v1);
    take(q2, v2);
  }
  for (int rb = lane + 2 * BSR_WAVE; rb < n_rb; rb += 2 * BSR_WAVE) {
    const bool two = rb + BSR_WAVE < n_rb;
    double q[BSR_P1_WORDS], q2[BSR_P1_WORDS];
    fetch(q, rb);
    fetch(q2, two ? rb + BSR_WAVE : rb);
    take(q, true);
    take(q2, two);
  }
  if constexpr (STAGED) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (lane + q * BSR_WAVE < CK_WORDS) sh_ck[lane + q * BSR_WAVE] = ckw[q];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
  }
  BSR_SOLVE_STAMP(t_s1);
  // the ten sums over the wave, eight and two at a time by half-wave and row swaps (bsr_device.h: wave_sum8 -- ~90 vector
  // instructions with the broadcasts instead of ~200 for ten butterflies of their own: these reductions were the largest
  // K-independent piece of the kernel after the loads, -DBSR_SOLVE_STAMPS).  A fixed order, as before -- another one.
  static_assert(BSR_NQ_MAX + 2 == 10, "eight sums and two");
  {
    auto lane_of = [](double v, int l) {
      return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    };
    double g[8], lo, hi;
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = sum[i];
    wave_sum8(g, lo, hi);                       // rows 0..3 of lo: totals 0, 2, 1, 3; of hi: 4, 6, 5, 7
    double a8 = sum[8], a9 = sum[9];
    swap32(a8, a9);
    a8 += a9;                                   // lanes 0-31: sum 8 over both halves; lanes 32-63: sum 9
    a8 = row_sum16(a8);
    a8 += dpp_f64<0x142, 0xA>(a8);              // rows 1 and 3 add their left neighbour: row 1 = sum 8, row 3 = sum 9
    sum[0] = lane_of(lo, 0);  sum[1] = lane_of(lo, 32); sum[2] = lane_of(lo, 16); sum[3] = lane_of(lo, 48);
    sum[4] = lane_of(hi, 0);  sum[5] = lane_of(hi, 32); sum[6] = lane_of(hi, 16); sum[7] = lane_of(hi, 48);
    sum[8] = lane_of(a8, 31); sum[9] = lane_of(a8, 63);
  }
  amax = wave_max(amax);
  fl = wave_or(fl);
  if (fl & BSR_F_INF) amax = INFINITY;
  BSR_SOLVE_STAMP(t_s2);

  PropCoef* cf = coef + p;
  bsr_score* out = outv + p;
  if (d_mode == BSR_MODE_EVAL) {
    if (lane == 0) {
      cf->skip = 1;
      out->maxabs = amax;
      out->flags = fl;
      out->rank = 0;
      out->loglik = out->sse = out->scale = out->smin = out->smax = 0.0;
      mhv[p].loglik = 0.0;
      mhv[p].rank = 0;
    }
    return;
  }
  const int K = d_K, k = d_k, nq = d_nq;
  const double s = d_s;
  const ChainB* ck = STAGED ? reinterpret_cast<const ChainB*>(sh_ck) : ck_dev;
  const uint32_t flags = fl | ck->flags_k[k];
  const double scale_ref = fmax(ck->m_other[k], amax);
  if (flags & (BSR_F_INF | BSR_F_NAN)) {  // matrix_rank: inf -> 0, NaN -> LinAlgError (reported as -1)
    if (lane == 0) {
      cf->skip = 1;
      out->loglik = NAN;
      out->sse = NAN;
      out->scale = (flags & BSR_F_NAN) ? NAN : INFINITY;
      out->maxabs = amax;
      out->smin = out->smax = NAN;
      out->rank = (flags & BSR_F_NAN) ? -1 : 0;
      out->flags = flags | BSR_F_RANKDEF;
      mhv[p].loglik = NAN;
      mhv[p].rank = out->rank;
    }
    if (lane < BSR_MAX_K) out->beta[lane] = NAN;
    return;
  }
  const double zz = sum[BSR_NQ_MAX], zy = sum[BSR_NQ_MAX + 1];
  if (K == 1) {
    // no sibling fixes the accumulation scale; ask the host to rescore with a matched prescale
    const double as = amax * s;
    if (as > 0.0 && (as > 0x1p400 || as < 0x1p-400)) {
      if (lane == 0) {
        cf->skip = 1;
        out->maxabs = amax;
        out->flags = flags | BSR_F_SCALE_RETRY;
        out->rank = 0;
        out->loglik = out->sse = NAN;
        mhv[p].loglik = NAN;
        mhv[p].rank = 0;
      }
      return;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < BSR_NQ_MAX; ++i) sh_c[i] = sum[i];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed (waves do not share sh_c)
  double cc = 0.0, cqy = 0.0;
#pragma unroll
  for (int i = 0; i < BSR_NQ_MAX; ++i) {
    if (i < nq) {
      cc = fma(sum[i], sum[i], cc);
      cqy = fma(sum[i], ck->qy[i], cqy);
    }
  }
  const double rho2 = zz - cc;
  // the candidate is (nearly) inside the span of the current columns: |w|^2 and w.y come from the direct residual pass
  const bool ambiguous = (nq > 0) && !(rho2 > 1e-6 * zz);
  // ... unless the host recognised the candidate as the current tree k itself (4-7 % of the real mix, three quarters of
  // what used to be flagged): then w = 0 exactly, which is what the residual pass would measure (|w|^2 ~ 1e-32 |s z|^2,
  // below the cut) -- the same arithmetic follows, without the pass.  The claim is only trusted when the one-pass
  // figure agrees that the candidate is in the span.
  const bool known_in_span = ambiguous && (d_dup & 1) != 0;
  if (ambiguous && !known_in_span) {
    if (lane < BSR_NQ_MAX) cf->c[lane] = (lane < nq) ? sh_c[lane] : 0.0;
    if (lane == 0) {
      cf->s = s;
      cf->zz = zz;
      cf->tau = 1.0 / (s * scale_ref);
      cf->scale = scale_ref;
      cf->maxabs = amax;
      cf->flags = flags;
      cf->skip = 0;
      // hand the proposal to the residual pass and k_finalize (the order of the list does not matter to the results)
      flagged[1 + atomicAdd(&flagged[0], 1)] = p;
    }
    return;
  }
  if (lane == 0) cf->skip = 1;
  SolveIn in;
  in.ck = ck;
  in.c = sh_c;
  in.rho2 = known_in_span ? 0.0 : rho2;
  in.zz = zz;
  in.wy = known_in_span ? 0.0 : zy - cqy;
  in.tau = 1.0 / (s * scale_ref);
  in.s = s;
  in.sigma = d_sigma;
  in.scale = scale_ref;
  in.maxabs = amax;
  in.K = K;
  in.k = k;
  in.N = N;
  in.flags = flags;
  in.rank_floor = rank_floor;
  in.exact = (d_dup & 2) ? 1 : 0;
  in.mh = mhv + p;
#ifdef BSR_SOLVE_STAMPS
  in.st[0] = t_s0; in.st[1] = t_s1; in.st[2] = t_s2; in.st[3] = 0;
#endif
  solve_any(in, lane, out);

}
