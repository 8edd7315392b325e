// HIP kernels of the MCMC-SymReg likelihood hot path for gfx950 (MI355X, wave64).
//
// Row pass (the O(N) work; one launch covers a whole batch of proposals):
//   k_rows<MODE_PROJECT>  : postfix stack-machine over each candidate tape (allcal, codes/funcs.py:175-220) fused
//                           with the projections of the candidate column on the chain's cached orthonormal basis,
//                           |z|^2, z.y, max|z| and the inf/NaN census.  Candidate columns are NOT stored: the few
//                           consumers (near-dependent candidates, an accepted proposal) re-run the tape instead.
//   k_rows<MODE_RESIDUAL> : same interpreter; direct residual of the candidate against the sibling basis
//                           (|w|^2, w.y) for the proposals k_solve could not settle from the projections.
// Per-proposal scalar work (K <= 8, one wave per proposal, lanes form an 8x8 grid):
//   k_solve    : reduce the partials, singular values of the K x K factor -> matrix_rank (codes/funcs.py:1226),
//                ridge OLS (codes/funcs.py:1151-1155), SSE and log-likelihood (codes/funcs.py:1162-1173).
//   k_finalize : the same algebra for the proposals that needed the residual pass.
// Rare path (initialisation / accepted proposal), one workgroup each:
//   k_refresh_basis : orthonormal basis of the chain's K current columns (Gram-Schmidt fallback).
//   k_chain_fit     : ridge OLS of y on the K current columns, with or without intercept
//                     (codes/funcs.py:1235 old state; codes/bsr_class.py:147-163, 211-233).
//
// Everything is deterministic: partial sums are written per (proposal,row block) and reduced in a fixed order.
// Compiled with -ffp-contract=off so that a*x+b keeps numpy's two roundings; accumulations use explicit fma().
#include <algorithm>
#include "bsr_internal.h"

#include <cstdlib>

#include "bsr_device.h"

// ---------------------------------------------------------------------------------------------------------------
// Row pass.  grid = 8 * ceil(n_rb/8) * n_pg workgroups of 4 waves (XCD-aware mapping above); a workgroup owns one
// row block of rb_rows rows and one group of pg proposals, a wave walks its share of the group's tapes over the
// block, 64*U rows per sweep.
//   LDS = true : the X columns referenced by the batch (feat_list) and y are staged once per workgroup in LDS;
//                terminals are ds_read_b128 (ids in the feature stream are LDS slots).
//   LDS = false: terminals and y come from global memory (L2); used when the referenced columns exceed the LDS
//                budget.
//   MODE_PROJECT : c = Q^T (s z), |s z|^2, (s z).y, max|z|, inf/NaN census -> part[(p,rb)][12]; optional z store
//                  (allcal / set_current).
//   MODE_RESIDUAL: w = s z - Q c for the proposals k_solve flagged; |w|^2 and w.y -> part[(p,rb)][2].
enum { MODE_PROJECT = 0, MODE_RESIDUAL = 1 };
int env_int(const char* name, int dflt);   // bsr_api.hip
#ifndef BSR_RESID_WGS
#define BSR_RESID_WGS 32u   // workgroups of the residual pass at most (16 waves each; see MODE_RESIDUAL in k_rows)
#endif
#ifndef BSR_SOLVE_WAVES
#define BSR_SOLVE_WAVES 4   // k_solve: proposals (waves) per workgroup (measured at C2 / K=8, us per step: 4: 17.3 / 27.2,
                           // 8: 17.9 / 29.7, 16: 19.7 / 37.7 -- tools/probes/lib_ab.sh; under direct dispatch, round 5:
                           // 1: 10.98 / 20.2, 2: 10.7-10.8 / 18.4, 4: 10.6 / 18.2, 8: 10.9-11.0 / 19.1, 16: 12.9 / 25.6;
                           // round 6, the leaner kernel: 1: 10.3 / 16.6, 2: 10.1 / 16.2, 4: 10.1-10.2 / 16.4, 8: 10.3 --
                           // profiles/r06_solve_waves_ab.txt; eight would also take 7 KB of LDS, more than a row-pass
                           // workgroup leaves free on its CU)
#endif

// In-kernel timing stamps (debug builds only: -DBSR_STAMPS): shader-clock samples of the first waves of the PROJECT
// pass, read back with bsr_debug_stamps().  tools/stamps.py prints the per-phase cycle budget of a wave.
#ifdef BSR_STAMPS
#define BSR_STAMP_WAVES 16384
#define BSR_STAMP_SLOTS 48
__device__ unsigned long long bsr_dbg_stamps[BSR_STAMP_WAVES * BSR_STAMP_SLOTS];
#define STAMP(i)                                                                                         \
  do {                                                                                                   \
    const int si_ = (i);                                                                                 \
    if (MODE == MODE_PROJECT && gwave < BSR_STAMP_WAVES && si_ < BSR_STAMP_SLOTS && lane == 0)           \
      bsr_dbg_stamps[gwave * BSR_STAMP_SLOTS + si_] = clock64();                                         \
  } while (0)
extern "C" int bsr_debug_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bsr_dbg_stamps), sizeof(unsigned long long) * n);
}
#else
#define STAMP(i) do {} while (0)
#endif

#include "bsr_solve.h"


// The residual pass (no LDS staging) runs FOUR of its 4-wave workgroups inside one 16-wave workgroup: a CU that hosts
// any of its waves cannot take a tile workgroup of the next batch, so what the pass costs the pipeline is CUs touched
// x time, and a quarter of the CUs do the same work in the same time (26 instead of 104 at N = 100k).
template <bool LDS, int MODE>
struct RowsShape {
  static constexpr bool fat = (MODE == MODE_RESIDUAL) && !LDS;
  static constexpr int waves = fat ? 4 * BSR_WG_WAVES : BSR_WG_WAVES;
  static constexpr int min_waves = fat ? 1 : BSR_ROWS_MIN_WAVES;
};
template <typename T, int NQ, int U, bool LDS, int MODE>
__global__ __launch_bounds__((RowsShape<LDS, MODE>::waves * BSR_WAVE), (RowsShape<LDS, MODE>::min_waves)) void k_rows(
    const T* __restrict__ Xt, const T* __restrict__ y, int64_t ld, int64_t N, const uint64_t* __restrict__ codes,
    const uint64_t* __restrict__ feats, const double* __restrict__ lnp, const PropDesc* __restrict__ desc,
    const PropCoef* __restrict__ coef, int P, int rb_rows, int pg, int n_rb, int n_pg,
    const int32_t* __restrict__ feat_list, int nF, double* __restrict__ part, T* __restrict__ spill,
    int spill_slots, int32_t* __restrict__ queue, int32_t* __restrict__ queue_clear, FinArgs fin) {
  constexpr bool DYN = !LDS && MODE == MODE_PROJECT;
  constexpr int S = (U >= 8) ? 2 : BSR_REG_STACK;
  constexpr int VEC = 16 / sizeof(T);
  extern __shared__ __align__(16) unsigned char smem[];
  T* sx = reinterpret_cast<T*>(smem);  // [nF][rb_rows] then y[rb_rows]   (LDS variant only)
  constexpr bool FAT = RowsShape<LDS, MODE>::fat;
  const int lane = threadIdx.x & 63;
  const int wave_raw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int vsub = FAT ? (wave_raw >> 2) : 0;      // which of the workgroup's four virtual 4-wave workgroups
  const int wave = FAT ? (wave_raw & 3) : wave_raw;
  const int vwg = FAT ? (int)blockIdx.x * 4 + vsub : (int)blockIdx.x;
  WorkItem wi = {0, 0, true};
  if constexpr (!DYN) {
    wi = map_work(n_rb, n_pg, vwg);
    if (!FAT && !wi.valid) return;
  }
  const PropDesc CONSTANT_AS* dsc = as_const(desc);
  const PropCoef CONSTANT_AS* cf = as_const(coef);
  // RESIDUAL: the proposals to redo are the ones k_solve appended to the batch's list (queue[0] = count, queue[1..] =
  // proposal indices); one workgroup per row block walks that list -- nothing to stage or schedule when it is empty
  const int32_t CONSTANT_AS* flagged = as_const(queue);
  const int n_flag = (MODE == MODE_RESIDUAL) ? flagged[0] : 0;
  constexpr bool FAT0 = RowsShape<LDS, MODE>::fat;
  // the finalise step can follow inside this kernel where its solver fits the residual pass's register budget (16 waves
  // per workgroup: 128 VGPRs; K >= 4 would spill)
  constexpr bool CAN_FUSE = FAT0 && NQ <= 3;
  const bool fused = CAN_FUSE && fin.ck != nullptr;
  if (MODE == MODE_RESIDUAL && n_flag == 0) return;   // nothing to do (the arrival counter stays zero)
  tables_to_lds();
  if (!LDS) __syncthreads();
  const bool active = !(FAT && !wi.valid);   // no early exit: the tail below has barriers
  if (LDS) {
    const int nvec = rb_rows / VEC;
    using V4 = __attribute__((ext_vector_type(4))) float;
    for (int idx = threadIdx.x; idx < (nF + 1) * nvec; idx += BSR_WG_WAVES * BSR_WAVE) {
      const int f = idx / nvec, v = idx - f * nvec;
      const T* src = (f < nF) ? Xt + (int64_t)feat_list[f] * ld : y;
      V4 val = {0.f, 0.f, 0.f, 0.f};
      if (src) val = *reinterpret_cast<const V4*>(src + (int64_t)wi.rb * rb_rows + (int64_t)v * VEC);
      *reinterpret_cast<V4*>(sx + (size_t)f * rb_rows + (size_t)v * VEC) = val;
    }
    __syncthreads();
  }
  const T* sy = sx + (size_t)nF * rb_rows;
  const int sweeps = rb_rows / (BSR_WAVE * U);
  const int gwave = vwg * BSR_WG_WAVES + wave;
  T* my_spill = spill ? spill + (size_t)gwave * spill_slots * (BSR_WAVE * 8) : nullptr;
  int stamp_i = 0;
  (void)stamp_i;
  STAMP(stamp_i++);  // wave start
#ifdef BSR_STAMPS
  if (MODE == MODE_PROJECT && gwave < BSR_STAMP_WAVES && lane == 0)
    bsr_dbg_stamps[gwave * BSR_STAMP_SLOTS + 46] = wall_clock64();  // constant-rate reference (100 MHz)
#endif

  // one task: tape p over row block rb (all its sweeps), partial sums stored for k_solve
  auto run_task = [&](const int p, const int rb, const int64_t row_base) {
    const T* qbase = (const T*)dsc[p].qbase;
    T* zout = (T*)dsc[p].zout;
    const double s = dsc[p].s;
    const uint64_t* pc = codes + dsc[p].code_off;
    const uint64_t* pf = feats + dsc[p].feat_off;
    const double* pl = lnp + 2 * (size_t)dsc[p].ln_off;
    const int n_nodes = dsc[p].n_nodes;

    double c[NQ > 0 ? NQ : 1];
#pragma unroll
    for (int i = 0; i < NQ; ++i) c[i] = (MODE == MODE_RESIDUAL) ? cf[p].c[i] : 0.0;
    double a0 = 0.0, a1 = 0.0, amax = 0.0;  // PROJECT: |s z|^2, s z.y   RESIDUAL: |w|^2, w.y
    STAMP(stamp_i++);  // proposal start (descriptor fields requested)

    for (int sw = 0; sw < sweeps; ++sw) {
      const int off = sw * (BSR_WAVE * U) + 2 * lane;  // lane's pair 0 inside the row block
      // sibling-basis values first: their latency hides under the tape
      T qv[NQ > 0 ? NQ : 1][U];
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const T* q = qbase + (int64_t)i * ld + row_base + off;
#pragma unroll
        for (int j = 0; j < U / 2; ++j) {
          qv[i][2 * j] = q[j * 128];
          qv[i][2 * j + 1] = q[j * 128 + 1];
        }
      }
      T yv[U];
      if (!LDS) {
#pragma unroll
        for (int j = 0; j < U / 2; ++j) {
          yv[2 * j] = y ? y[row_base + off + j * 128] : (T)0;
          yv[2 * j + 1] = y ? y[row_base + off + j * 128 + 1] : (T)0;
        }
      }
      T acc[U];
      if (LDS) {
        LdsCols<T, U> ldr{sx, rb_rows, off};
        run_tape<T, U, S>(pc, pf, pl, n_nodes, ldr, acc, my_spill, lane);
#pragma unroll
        for (int j = 0; j < U / 2; ++j) {
          yv[2 * j] = sy[off + j * 128];
          yv[2 * j + 1] = sy[off + j * 128 + 1];
        }
      } else {
        GlobalCols<T, U> ldr{Xt, ld, row_base + off};
        run_tape<T, U, S>(pc, pf, pl, n_nodes, ldr, acc, my_spill, lane);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = off + (u >> 1) * 128 + (u & 1);
        const bool valid = (row_base + r) < N;
        const T z = valid ? acc[u] : (T)0;
        acc[u] = z;
        const double zd = (double)z;
        const double zs = zd * s;
        if (MODE == MODE_PROJECT) {
          amax = max_abs(amax, zd);
          a0 = fma(zs, zs, a0);
          a1 = fma(zs, (double)yv[u], a1);
#pragma unroll
          for (int i = 0; i < NQ; ++i) c[i] = fma((double)qv[i][u], zs, c[i]);
        } else {
          double w = zs;
#pragma unroll
          for (int i = 0; i < NQ; ++i) w = fma(-c[i], (double)qv[i][u], w);
          if (valid) {
            a0 = fma(w, w, a0);
            a1 = fma(w, (double)yv[u], a1);
          }
        }
      }
      STAMP(stamp_i++);  // sweep done
      if (MODE == MODE_PROJECT && zout) {
        T* zo = zout + row_base + off;
#pragma unroll
        for (int j = 0; j < U / 2; ++j) {
          zo[j * 128] = acc[2 * j];
          zo[j * 128 + 1] = acc[2 * j + 1];
        }
      }
    }
    if (MODE == MODE_PROJECT) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) c[i] = wave_sum(c[i]);
      a0 = wave_sum(a0);
      a1 = wave_sum(a1);
      amax = wave_max(amax);
      // census after the fact: max|z| is +inf iff an inf is present (fmax ignores NaN); the sum of squares is
      // NaN iff a NaN is present (squares are non-negative, so no inf-inf)
      const uint32_t fl = ((amax == INFINITY) ? BSR_F_INF : 0u) | (isnan(a0) ? BSR_F_NAN : 0u);
      STAMP(stamp_i++);  // reductions done
      // every lane holds the wave totals and stores them to the same addresses: the task body has no lane-divergent
      // branch, so its uniform node loop is left alone by the CFG structurizer
      {
        double* o = part + ((size_t)p * n_rb + rb) * BSR_P1_WORDS;
#pragma unroll
        for (int i = 0; i < BSR_NQ_MAX; ++i) o[i] = (i < NQ) ? c[i < NQ ? i : 0] : 0.0;
        o[8] = a0;
        o[9] = a1;
        o[10] = amax;
        o[11] = (double)fl;
      }
    } else {
      a0 = wave_sum(a0);
      a1 = wave_sum(a1);
      {
        double* o = part + ((size_t)p * n_rb + rb) * BSR_P2_WORDS;
        o[0] = a0;
        o[1] = a1;
      }
    }
  };

  if constexpr (DYN) {
    // Work queue: every workgroup pulls (four tapes, row block) tickets until none are left.  Tapes differ several-fold in cost, a
    // launch is only 2-3 tasks per resident wave, and the SIMD favours its oldest waves, so a static split leaves the
    // chip waiting for a few late, lonely waves (measured with tools/stamps.py: waves of one launch end anywhere
    // between 6 and 28 us).  Here the heaviest tapes go first (desc[i].order) and the tail is one task long.
    //  * Ticket counters are per XCD (blockIdx & 7, the hardware's round-robin placement): an XCD works on its own
    //    eighth of the row blocks, like the static mapping, and bumps its counters with agent-scope relaxed atomics.
    //    hipcc (ROCm 7.2, gfx950) emits `global_atomic_add ... sc0` for them -- the same encoding as for workgroup
    //    scope, executed in the XCD's L2; only system scope adds sc1 (memory-side, ~35 ns each once thousands of waves
    //    hit one address).  Should a counter ever be touched from two XCDs, each L2 counts through its own copy: tasks
    //    may be repeated (the same values stored again), never skipped.
    //  * Even in L2 one address takes an atomic only every ~8 ns, so an XCD's tasks are spread over BSR_QUEUE_SUB
    //    counters (row block j of the XCD belongs to counter j % BSR_QUEUE_SUB).  A workgroup drains its own counter,
    //    then reads all of them at once (one lane each, through the atomic unit so the values are fresh) and moves to
    //    one that still has tickets; which one depends on the workgroup, so the last counters are not stormed.
    //  * A ticket is one row block x four neighbouring tapes of the cost order, one per wave: the four waves of a
    //    workgroup stay on one row block (its y, basis and X lines are shared through the CU's L1, as in the static
    //    mapping; per-wave tickets scattered them and made every sweep 1.8x slower) and finish at about the same time.
    //  * The next ticket is requested before the current task runs, so its round trip hides under the task.
    //  * Counters live 128 B apart; the launcher hands out a fresh, zeroed set per launch and this launch clears the
    //    set that comes up again half a ring later.
    const int x = blockIdx.x & 7;
    if (blockIdx.x == 0 && threadIdx.x < 8 * BSR_QUEUE_SUB)
      __hip_atomic_store(&queue_clear[threadIdx.x * 32], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ int s_tk[2], s_next;
    const int nrb_x = (n_rb - x + 7) >> 3;
    int32_t* qx = queue + x * (BSR_QUEUE_SUB * 32);
    const int wgid = (int)(blockIdx.x >> 3);
    const int nquad = (P + BSR_WG_WAVES - 1) / BSR_WG_WAVES;  // a ticket = one row block x four neighbouring tapes
    const int my_c = lane & (BSR_QUEUE_SUB - 1);
    const int my_n = ((nrb_x - my_c + BSR_QUEUE_SUB - 1) / BSR_QUEUE_SUB) * nquad;  // tickets behind counter my_c
    int cq = wgid & (BSR_QUEUE_SUB - 1);
    int par = 0;
    for (int round = 0;; ++round) {
      const int nj = (nrb_x - cq + BSR_QUEUE_SUB - 1) / BSR_QUEUE_SUB;  // row blocks cq, cq+SUB, ... of this XCD
      const int n_tasks = nj * nquad;
      if (n_tasks > 0) {
        int32_t* q = qx + cq * 32;
        if (threadIdx.x == 0) s_tk[par] = __hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        for (;;) {
          const int t = __builtin_amdgcn_readfirstlane(s_tk[par]);
          if (t >= n_tasks) break;
          int tn = 0;
          if (threadIdx.x == 0) tn = __hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // row-block-major: neighbouring tickets are different tapes on one row block (tape-major order put every wave
          // of the chip on the same two or three X columns at once and doubled the sweep time)
          const int jj = t / nquad;
          const int pi = (t - jj * nquad) * BSR_WG_WAVES + wave;
          const int rb = x + 8 * (cq + BSR_QUEUE_SUB * jj);
          if (pi < P) run_task(dsc[pi].order, rb, (int64_t)rb * rb_rows);
          if (threadIdx.x == 0) s_tk[par ^ 1] = tn;
          par ^= 1;
          __syncthreads();
        }
      }
      if (wave == 0) {
        int seen = 0x7fffffff;
        if (lane < BSR_QUEUE_SUB)
          seen = __hip_atomic_fetch_add(qx + my_c * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t open = (uint32_t)__ballot(lane < BSR_QUEUE_SUB && seen < my_n);
        // rotate by a workgroup- and round-dependent amount and take the first open counter from there
        const int rot = (wgid * 5 + round * 7) & (BSR_QUEUE_SUB - 1);
        const uint32_t m16 = (1u << BSR_QUEUE_SUB) - 1u;
        const uint32_t r = ((open >> rot) | (open << (BSR_QUEUE_SUB - rot))) & m16;
        if (lane == 0) s_next = open ? ((rot + __builtin_ctz(r | 0x10000u)) & (BSR_QUEUE_SUB - 1)) : -1;
      }
      __syncthreads();
      cq = __builtin_amdgcn_readfirstlane(s_next);
      if (cq < 0) break;
    }
  } else {
    if (MODE == MODE_RESIDUAL) {
      if constexpr (FAT) {
        // the launch is capped at BSR_RESID_WGS workgroups (an empty list -- nineteen batches in twenty -- then costs a few
        // microseconds at any N instead of 0.1 us per row block's workgroup: 25 us at N = 1M); its virtual workgroups
        // stride over the row blocks
        // ... and (row block, flagged proposal) pairs are dealt to ALL waves of the launch, a row block's proposals to
        // neighbouring waves: one flagged proposal keeps every wave busy, not one wave in four
        const int n_vwg = ((n_rb + 7) / 8) * 8;
        const int n_units = n_vwg * n_flag;
        const int n_waves = (int)gridDim.x * 4 * BSR_WG_WAVES;
        for (int u = (int)blockIdx.x * 4 * BSR_WG_WAVES + wave_raw; u < n_units; u += n_waves) {
          const int vw = u / n_flag, pi = u - vw * n_flag;
          const WorkItem w2 = map_work(n_rb, n_pg, vw);
          if (w2.valid) run_task(flagged[1 + pi], w2.rb, (int64_t)w2.rb * rb_rows);
        }
      } else if (active) {
        for (int pi = wave; pi < n_flag; pi += BSR_WG_WAVES) run_task(flagged[1 + pi], wi.rb, (int64_t)wi.rb * rb_rows);
      }
    } else {
      for (int pi = wave; pi < pg; pi += BSR_WG_WAVES) {
        const int p = wi.pgi * pg + pi;
        if (p >= P) break;
        run_task(p, wi.rb, (int64_t)wi.rb * rb_rows);
      }
    }
  }
#ifdef BSR_STAMPS
  if (MODE == MODE_PROJECT && gwave < BSR_STAMP_WAVES && lane == 0)
    bsr_dbg_stamps[gwave * BSR_STAMP_SLOTS + 47] = wall_clock64();
#endif
  if constexpr (CAN_FUSE) {
    if (!fused) return;
    __shared__ int s_last;
    __shared__ double sh_fin[4 * BSR_WG_WAVES][BSR_NQ_MAX];
    double* sh_c = sh_fin[wave_raw];
    // one flagged proposal: its residual sums over the row blocks (one wave, blocks lane-strided, as k_finalize), then
    // the solve with the exact residual
    auto finalize_one = [&](const int p) {
      const PropCoef* cfp = coef + p;
      double ww = 0.0, wy = 0.0;
      for (int rb = lane; rb < n_rb; rb += BSR_WAVE) {
        const double* q = part + ((size_t)p * n_rb + rb) * BSR_P2_WORDS;
        ww += __builtin_nontemporal_load(q);        // written by other waves of this launch: not through a
        wy += __builtin_nontemporal_load(q + 1);    // possibly stale line of this CU's caches
      }
      ww = wave_sum(ww);
      wy = wave_sum(wy);
      if (lane < BSR_NQ_MAX) sh_c[lane] = cfp->c[lane];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
      SolveIn in;
      in.ck = fin.ck + dsc[p].ck;
      in.c = sh_c;
      in.rho2 = ww;
      in.zz = cfp->zz;
      in.wy = wy;
      in.tau = cfp->tau;
      in.s = cfp->s;
      in.sigma = dsc[p].sigma;
      in.scale = cfp->scale;
      in.maxabs = cfp->maxabs;
      in.K = dsc[p].K;
      in.k = dsc[p].k;
      in.N = N;
      in.flags = cfp->flags;
      in.rank_floor = fin.rank_floor;
      in.exact = (dsc[p].self_dup & 2) ? 1 : 0;
      in.mh = fin.mh + p;
      // the same tiers as k_solve / k_finalize (solve_any): a proposal's bytes must not depend on the route it took
      // (the span shortcut settles in k_solve what the residual route settles here: tests compare the two byte for byte).
      // The fast tier's registers on top of this kernel's exceed its 128: a few spills in this tail, run by the launch's
      // last workgroup for a handful of proposals (tests/test_build_resources.py bounds them)
      if constexpr (NQ >= 1) {
        if (!in.exact && solve_fast<NQ>(in, lane, fin.out + p)) return;
      }
      if constexpr (NQ >= 1 && NQ <= 4) solve_regs<NQ>(in, lane, fin.out + p);
      else if constexpr (NQ >= 5) solve_cols<NQ>(in, lane, fin.out + p);
    };
    // Finalise, fused: every workgroup publishes its residual sums and checks in; the last one to arrive runs the
    // flagged proposals' solves on its sixteen waves (one launch, one launch gap and one blocked CU fewer per batch).
    // records and counter are in uncached memory: a store is globally visible once it is acknowledged, a load never
    // sees a cache -- waiting for the wave's own stores is the whole release, and there is nothing to acquire
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const int old = __hip_atomic_fetch_add(fin.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (old == (int)gridDim.x - 1) ? 1 : 0;
      if (s_last) __hip_atomic_store(fin.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    for (int fi = wave_raw; fi < n_flag; fi += 4 * BSR_WG_WAVES) finalize_one(flagged[1 + fi]);
  }
}


__global__ __launch_bounds__(BSR_SOLVE_WAVES * BSR_WAVE) void k_solve(const PropDesc* __restrict__ desc, const ChainB* __restrict__ cks,
                                                    int P, int n_rb, const double* __restrict__ part1, int64_t N,
                                                    PropCoef* __restrict__ coef, bsr_score* __restrict__ outv,
                                                    double rank_floor, int32_t* __restrict__ flagged,
                                                    MhRes* __restrict__ mhv, int32_t* __restrict__ flagged_next) {
  // the slot's two lists of flagged proposals alternate between batches: this batch appends to `flagged`, the
  // residual pass and k_finalize read it, and the other list -- consumed by the batch before -- is emptied here
  if (blockIdx.x == 0 && threadIdx.x == 0 && flagged_next) flagged_next[0] = 0;
  // one wave per proposal, four waves per workgroup: a quarter of the CUs touched (a tile workgroup of another batch
  // cannot start on a CU that hosts even one of these waves); eight or sixteen per workgroup measured slower -- the
  // bigger workgroup itself waits longer for a CU with that many registers free
  const int wave_id = threadIdx.x >> 6;
  // (the same in every lane of the wave, and said so: the proposal's descriptor then comes in by scalar loads, whole, with
  // the first field asked for -- as vector loads its fields arrived one dependent round trip after the other)
  const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * BSR_SOLVE_WAVES + wave_id);
  if (p >= P) return;
  const int lane = threadIdx.x & 63;
  const PropDesc CONSTANT_AS* dsc = as_const(desc);
  __shared__ double sh_all_c[BSR_SOLVE_WAVES][BSR_NQ_MAX];
  __shared__ double sh_all_ck[BSR_SOLVE_WAVES][BSR_SOLVE_CK_WORDS];   // the wave's copy of its chain's block (bsr_solve.h): 3.6 KB with sh_all_c
  double* sh_c = sh_all_c[wave_id];

  solve_proposal<false, true>(dsc, cks, p, n_rb, part1, N, coef, outv, rank_floor, flagged, mhv, lane, sh_c, sh_all_ck[wave_id]);
}

// ---------------------------------------------------------------------------------------------------------------
// finalize (only proposals flagged by k_solve): same algebra with the directly measured |w|^2 and w.y
__global__ __launch_bounds__(4 * BSR_WAVE) void k_finalize(const PropDesc* __restrict__ desc,
                                                           const ChainB* __restrict__ cks,
                                                           const PropCoef* __restrict__ coef, int P, int n_rb,
                                                           const double* __restrict__ part2, int64_t N,
                                                           bsr_score* __restrict__ outv, double rank_floor,
                                                           int32_t* __restrict__ flagged, MhRes* __restrict__ mhv) {
  // the waves of all workgroups walk the batch's list of flagged proposals, one proposal per wave at a time (the
  // list is emptied by the next batch's k_solve)
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const PropDesc CONSTANT_AS* dsc = as_const(desc);
  __shared__ double sh_all[4][BSR_NQ_MAX];
  double* sh_c = sh_all[wave];
  const int n_flag = flagged[0];
  for (int fi = blockIdx.x * 4 + wave; fi < n_flag; fi += gridDim.x * 4) {
  const int p = flagged[1 + fi];
  const PropCoef* cf = coef + p;
  double ww = 0.0, wy = 0.0;
  for (int rb = lane; rb < n_rb; rb += BSR_WAVE) {
    const double* q = part2 + ((size_t)p * n_rb + rb) * BSR_P2_WORDS;
    ww += q[0];
    wy += q[1];
  }
  ww = wave_sum(ww);
  wy = wave_sum(wy);
  if (lane < BSR_NQ_MAX) sh_c[lane] = cf->c[lane];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
  SolveIn in;
  in.ck = cks + dsc[p].ck;
  in.c = sh_c;
  in.rho2 = ww;
  in.zz = cf->zz;
  in.wy = wy;
  in.tau = cf->tau;
  in.s = cf->s;
  in.sigma = dsc[p].sigma;
  in.scale = cf->scale;
  in.maxabs = cf->maxabs;
  in.K = dsc[p].K;
  in.k = dsc[p].k;
  in.N = N;
  in.flags = cf->flags;
  in.rank_floor = rank_floor;
  in.exact = (dsc[p].self_dup & 2) ? 1 : 0;
  in.mh = mhv + p;
  solve_any(in, lane, outv + p);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The scalar tail of newProp on the device (codes/funcs.py:1226-1306): log-ratio, accept test and, per chain, the
// first proposal of the speculative run that is not "rejected as speculated".  One lane per chain span; the terms
// come from the host (include/bsr_hip.h), the additions are made in the reference's order, so logR -- and the
// decision -- are bit-identical to the host's.
__global__ __launch_bounds__(BSR_WAVE) void k_events(const MhRes* __restrict__ mh, const double* __restrict__ terms,
                                                     const int32_t* __restrict__ flags,
                                                     const int32_t* __restrict__ span_off, int n_spans, int K,
                                                     bsr_event* __restrict__ events) {
  for (int sp = blockIdx.x * BSR_WAVE + threadIdx.x; sp < n_spans; sp += gridDim.x * BSR_WAVE) {
    const int lo = span_off[sp], hi = span_off[sp + 1];
    bsr_event ev;
    ev.index = hi - lo;
    ev.kind = BSR_EV_NONE;
    ev.logR = NAN;
    for (int i = lo; i < hi; ++i) {
      const bool no_u = (flags[i] & BSR_MH_NO_UNIFORM) != 0;
      if (mh[i].rank < K) {                       // rank gate (NaN candidates, rank -1, come this way too)
        if (no_u) continue;                       // speculated: the run goes on
        ev.index = i - lo;
        ev.kind = BSR_EV_GATE;
        break;
      }
      const double* t = terms + 8 * (size_t)i;
      const double log_y = mh[i].loglik - t[0];
      double logR;
      if (flags[i] & BSR_MH_JUMP) logR = log_y + t[1] + t[2] + t[3] + t[4];
      else logR = log_y + t[1] + t[2];
      logR = logR + t[5] - t[6];
      if (no_u) {                                 // passed the gate against the speculation: the host draws the uniform
        ev.index = i - lo;
        ev.kind = BSR_EV_GATE_PASSED;
        ev.logR = logR;
        break;
      }
      const double alpha = (0 < logR) ? 0 : logR;  // Python's min(logR, 0)
      if (!(t[7] >= alpha)) {
        ev.index = i - lo;
        ev.kind = BSR_EV_ACCEPT;
        ev.logR = logR;
        break;
      }
    }
    events[sp] = ev;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// rare path: single-workgroup kernels (1024 threads = 16 waves)
#define BSR_BLK 1024
#define BSR_BLK_WAVES (BSR_BLK / BSR_WAVE)

template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sh) {  // sh: (BSR_BLK_WAVES+1)*NV doubles
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sh[w * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double t = 0.0;
    for (int ww = 0; ww < BSR_BLK_WAVES; ++ww) t += sh[ww * NV + threadIdx.x];
    sh[BSR_BLK_WAVES * NV + threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = sh[BSR_BLK_WAVES * NV + i];
}

__device__ __forceinline__ double pow2_prescale(double m) {
  if (!(m > 0.0) || isinf(m)) return 1.0;
  int e;
  frexp(m, &e);
  e = max(-1000, min(1000, e));
  return ldexp(1.0, -e);
}

// The chain's basis by Gram-Schmidt (fallback of the Cholesky-QR pipeline in bsr_refresh.hip: dependent or
// non-finite columns).  grid = 1 block of 1024 threads.  O_m d_m = Q R over all K current columns, three
// orthogonalisation sweeps per column; a column that holds inf/NaN enters as a zero column, a column that is exactly
// dependent on its predecessors ends as a zero basis vector with R_mm = 0.
template <typename T>
__global__ __launch_bounds__(BSR_BLK) void k_refresh_basis(const T* __restrict__ cur, T* __restrict__ Q,
                                                           const T* __restrict__ y, int64_t ld, int64_t N, int K,
                                                           const double* __restrict__ col_maxabs,
                                                           const uint32_t* __restrict__ col_flags,
                                                           ChainB* __restrict__ cb) {
  __shared__ double sh[(BSR_BLK_WAVES + 1) * 9];
  __shared__ double shR[BSR_NQ_MAX * BSR_NQ_MAX];
  __shared__ double shqy[BSR_NQ_MAX];
  __shared__ double shd[BSR_NQ_MAX];

  if (threadIdx.x < BSR_NQ_MAX * BSR_NQ_MAX) shR[threadIdx.x] = 0.0;
  if (threadIdx.x < BSR_NQ_MAX) {
    shqy[threadIdx.x] = 0.0;
    const int j = threadIdx.x;
    shd[j] = (j < K && !(col_flags[j] & (BSR_F_INF | BSR_F_NAN))) ? pow2_prescale(col_maxabs[j]) : 0.0;
    // per-k census of the sibling columns
    double m_other = 0.0;
    uint32_t fl = 0;
    for (int i = 0; i < K; ++i) {
      if (i == j) continue;
      m_other = fmax(m_other, col_maxabs[i]);
      fl |= col_flags[i] & (BSR_F_INF | BSR_F_NAN);
    }
    if (fl & BSR_F_INF) m_other = INFINITY;
    cb->s_k[j] = (j < K) ? pow2_prescale(m_other) : 1.0;
    cb->m_other[j] = (j < K) ? m_other : 0.0;
    cb->flags_k[j] = (j < K) ? fl : 0u;
  }
  __syncthreads();

  for (int m = 0; m < K; ++m) {
    const T* src = cur + (int64_t)m * ld;
    T* v = Q + (int64_t)m * ld;
    const double dm = shd[m];
    double h[9];
    // sweep 0: v = d_m * O_m ; h = Q^T v
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += BSR_BLK) {
      const double x = (dm != 0.0) ? dm * (double)src[n] : 0.0;
      v[n] = (T)x;
      const double xv = (double)v[n];
#pragma unroll
      for (int i = 0; i < BSR_NQ_MAX; ++i)
        if (i < m) h[i] = fma((double)Q[(int64_t)i * ld + n], xv, h[i]);
      h[8] = fma(xv, xv, h[8]);
    }
    block_sum<9>(h, sh);
    // three orthogonalisation sweeps: v -= Q h ; h = Q^T v
    for (int sweep = 0; sweep < 3; ++sweep) {
      if (m == 0) break;
      if (threadIdx.x < BSR_NQ_MAX && (int)threadIdx.x < m) shR[threadIdx.x * BSR_NQ_MAX + m] += h[threadIdx.x];
      double hn[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) hn[i] = 0.0;
      for (int64_t n = threadIdx.x; n < N; n += BSR_BLK) {
        double x = (double)v[n];
#pragma unroll
        for (int i = 0; i < BSR_NQ_MAX; ++i)
          if (i < m) x = fma(-h[i], (double)Q[(int64_t)i * ld + n], x);
        v[n] = (T)x;
        const double xv = (double)v[n];
#pragma unroll
        for (int i = 0; i < BSR_NQ_MAX; ++i)
          if (i < m) hn[i] = fma((double)Q[(int64_t)i * ld + n], xv, hn[i]);
        hn[8] = fma(xv, xv, hn[8]);
      }
      block_sum<9>(hn, sh);
#pragma unroll
      for (int i = 0; i < 9; ++i) h[i] = hn[i];
    }
    const double r = sqrt(h[8]);
    const double inv = (r > 0.0) ? 1.0 / r : 0.0;
    double qy[1] = {0.0};
    for (int64_t n = threadIdx.x; n < N; n += BSR_BLK) {
      const double x = (double)v[n] * inv;
      v[n] = (T)x;
      qy[0] = fma((double)v[n], (double)y[n], qy[0]);
    }
    block_sum<1>(qy, sh);
    if (threadIdx.x == 0) {
      shR[m * BSR_NQ_MAX + m] = r;
      shqy[m] = qy[0];
    }
    __syncthreads();
  }
  __syncthreads();
  // |y - Q Q^T y|^2 measured directly
  double yp[1] = {0.0};
  for (int64_t n = threadIdx.x; n < N; n += BSR_BLK) {
    double r = (double)y[n];
#pragma unroll
    for (int i = 0; i < BSR_NQ_MAX; ++i)
      if (i < K) r = fma(-shqy[i], (double)Q[(int64_t)i * ld + n], r);
    yp[0] = fma(r, r, yp[0]);
  }
  block_sum<1>(yp, sh);
  if (threadIdx.x < BSR_NQ_MAX * BSR_NQ_MAX) cb->R[threadIdx.x] = shR[threadIdx.x];
  if (threadIdx.x < BSR_NQ_MAX) {
    cb->qy[threadIdx.x] = shqy[threadIdx.x];
    cb->d[threadIdx.x] = shd[threadIdx.x];
  }
  if (threadIdx.x == 0) cb->yperp2 = yp[0];
}

// Ridge OLS of y on [1?, O_0..O_{K-1}] exactly as codes/funcs.py:1148-1162 / codes/bsr_class.py:147-163.
#define BSR_FIT_M (BSR_MAX_K + 1)
template <typename T>
__global__ __launch_bounds__(BSR_BLK) void k_chain_fit(const T* __restrict__ cols, const T* __restrict__ y,
                                                       int64_t ld, int64_t N, int K, int icpt,
                                                       ChainFitOut* __restrict__ out) {
  const int M = K + icpt;
  __shared__ double sh[(BSR_BLK_WAVES + 1) * 8];
  __shared__ double shA[BSR_FIT_M * BSR_FIT_M], shI[BSR_FIT_M * BSR_FIT_M], shg[BSR_FIT_M], shb[BSR_FIT_M];
  __shared__ double shmax[BSR_MAX_K];
  __shared__ uint32_t shfl[BSR_MAX_K];
  __shared__ int shpiv;

  // sweep A: max |.| and inf/NaN census per column
  for (int j = 0; j < K; ++j) {
    double mx = 0.0;
    uint32_t fl = 0;
    const T* c = cols + (int64_t)j * ld;
    for (int64_t n = threadIdx.x; n < N; n += BSR_BLK) {
      const double v = (double)c[n];
      if (isinf(v)) fl |= BSR_F_INF;
      if (isnan(v)) fl |= BSR_F_NAN;
      mx = fmax(mx, fabs(v));
    }
    mx = wave_max(mx);
    fl = wave_or(fl);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      sh[threadIdx.x >> 6] = mx;
      sh[BSR_BLK_WAVES + (threadIdx.x >> 6)] = (double)fl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double m2 = 0.0;
      uint32_t f2 = 0;
      for (int w = 0; w < BSR_BLK_WAVES; ++w) {
        m2 = fmax(m2, sh[w]);
        f2 |= (uint32_t)sh[BSR_BLK_WAVES + w];
      }
      if (f2 & BSR_F_INF) m2 = INFINITY;
      shmax[j] = m2;
      shfl[j] = f2;
    }
    __syncthreads();
  }
  double scale = icpt ? 1.0 : 0.0;
  uint32_t anyfl = 0;
  for (int j = 0; j < K; ++j) {
    scale = fmax(scale, shmax[j]);
    anyfl |= shfl[j];
  }
  if (threadIdx.x < K) {
    out->maxabs[threadIdx.x] = shmax[threadIdx.x];
    out->colflags[threadIdx.x] = shfl[threadIdx.x];
  }
  if (anyfl & (BSR_F_INF | BSR_F_NAN)) {
    if (threadIdx.x == 0) {
      out->sse = NAN;
      out->scale = (anyfl & BSR_F_NAN) ? NAN : INFINITY;
      out->anyflags = anyfl;
    }
    if (threadIdx.x < BSR_FIT_M) {
      out->beta[threadIdx.x] = NAN;
      out->beta_unscaled[threadIdx.x] = NAN;
    }
    return;
  }
  const double s = pow2_prescale(scale);
  const double tau = 1.0 / (s * scale);

  // sweep B: Gram of the prescaled columns and X^T y, one (i,j) pair group at a time
  for (int i = 0; i < M; ++i) {
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    // columns j = i..M-1 handled in chunks of 7, slot 7 = X_i . y
    for (int j0 = i; j0 < M; j0 += 7) {
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = 0.0;
      for (int64_t n = threadIdx.x; n < N; n += BSR_BLK) {
        const double vi = (icpt && i == 0) ? s : s * (double)cols[(int64_t)(i - icpt) * ld + n];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const int j = j0 + q;
          if (j < M) {
            const double vj = (icpt && j == 0) ? s : s * (double)cols[(int64_t)(j - icpt) * ld + n];
            acc[q] = fma(vi, vj, acc[q]);
          }
        }
        if (j0 == i) acc[7] = fma(vi, (double)y[n], acc[7]);
      }
      block_sum<8>(acc, sh);
      if (threadIdx.x == 0) {
        for (int q = 0; q < 7; ++q) {
          const int j = j0 + q;
          if (j < M) {
            const double a = tau * (tau * acc[q]);
            shA[i * BSR_FIT_M + j] = a;
            shA[j * BSR_FIT_M + i] = a;
          }
        }
        if (j0 == i) shg[i] = tau * acc[7];
      }
      __syncthreads();
    }
  }
  if (threadIdx.x < BSR_FIT_M * BSR_FIT_M) {
    const int i = threadIdx.x / BSR_FIT_M, j = threadIdx.x % BSR_FIT_M;
    if (i < M && j < M) {
      if (i == j) shA[threadIdx.x] += 1e-6;
      shI[threadIdx.x] = (i == j) ? 1.0 : 0.0;
    } else {
      shA[threadIdx.x] = (i == j) ? 1.0 : 0.0;
      shI[threadIdx.x] = (i == j) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  // Gauss-Jordan with partial pivoting, one thread per matrix element
  for (int col = 0; col < M; ++col) {
    if (threadIdx.x == 0) {
      double best = -1.0;
      int piv = col;
      for (int r = col; r < M; ++r) {
        const double v = fabs(shA[r * BSR_FIT_M + col]);
        if (v > best) { best = v; piv = r; }
      }
      shpiv = piv;
    }
    __syncthreads();
    const int piv = shpiv;
    double a = 0.0, b = 0.0;
    const int i = threadIdx.x / BSR_FIT_M, j = threadIdx.x % BSR_FIT_M;
    const bool act = threadIdx.x < BSR_FIT_M * BSR_FIT_M;
    if (act) {
      const int si = (i == col) ? piv : ((i == piv) ? col : i);  // row after the swap
      const double d = shA[piv * BSR_FIT_M + col];
      const double rowA = shA[piv * BSR_FIT_M + j], rowI = shI[piv * BSR_FIT_M + j];
      if (i == col) {
        a = rowA / d;
        b = rowI / d;
      } else {
        const double f = shA[si * BSR_FIT_M + col] / d;
        a = shA[si * BSR_FIT_M + j] - f * rowA;
        b = shI[si * BSR_FIT_M + j] - f * rowI;
      }
    }
    __syncthreads();
    if (act) {
      shA[threadIdx.x] = a;
      shI[threadIdx.x] = b;
    }
    __syncthreads();
  }
  if (threadIdx.x < BSR_FIT_M) {
    double t = 0.0;
    for (int j = 0; j < M; ++j) t += shI[threadIdx.x * BSR_FIT_M + j] * shg[j];
    shb[threadIdx.x] = (threadIdx.x < (unsigned)M) ? t : 0.0;
  }
  __syncthreads();
  // sweep C: direct residual
  double sse[1] = {0.0};
  for (int64_t n = threadIdx.x; n < N; n += BSR_BLK) {
    double fit = icpt ? shb[0] * (tau * s) : 0.0;
    for (int j = 0; j < K; ++j) fit = fma(shb[j + icpt] * (tau * s), (double)cols[(int64_t)j * ld + n], fit);
    const double r = (double)y[n] - fit;
    sse[0] = fma(r, r, sse[0]);
  }
  block_sum<1>(sse, sh);
  if (threadIdx.x == 0) {
    out->sse = sse[0];
    out->scale = scale;
    out->anyflags = anyfl;
  }
  if (threadIdx.x < BSR_FIT_M) {
    out->beta[threadIdx.x] = shb[threadIdx.x];
    out->beta_unscaled[threadIdx.x] = shb[threadIdx.x] / scale;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// layout helpers
template <typename T>
__global__ void k_transpose_in(const double* __restrict__ src, T* __restrict__ dst, int64_t N, int d, int64_t ld) {
  // 64 rows x d features per workgroup through LDS so both sides stay coalesced
  extern __shared__ double tile[];
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  const int rows = (int)min((int64_t)64, N - n0);
  for (int idx = threadIdx.x; idx < rows * d; idx += blockDim.x) tile[idx] = src[n0 * d + idx];
  __syncthreads();
  for (int idx = threadIdx.x; idx < d * 64; idx += blockDim.x) {
    const int f = idx >> 6, r = idx & 63;
    if (r < rows) dst[(int64_t)f * ld + n0 + r] = (T)tile[r * d + f];
  }
}
template <typename T>
__global__ void k_convert_out(const T* __restrict__ src, double* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (double)src[i];
}
template <typename T>
__global__ void k_convert_in(const double* __restrict__ src, T* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (T)src[i];
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
template <typename T, int NQ, int MODE, int U>
static void launch_rows_u(hipStream_t st, const RowPassArgs<T>& a) {
  const LaunchGeom& g = a.g;
  dim3 grid((unsigned)(((g.n_rb + 7) / 8) * 8 * g.n_pg)), block(BSR_WG_WAVES * BSR_WAVE);
  if (!a.feat_list && MODE == MODE_PROJECT) grid.x = (unsigned)g.dyn_wgs;  // work-queue launch
  if (MODE == MODE_RESIDUAL) grid.x = (unsigned)(((g.n_rb + 7) / 8) * 8);   // one (virtual) workgroup per row block, flagged list
  if (MODE == MODE_RESIDUAL && !a.feat_list) {   // four of them per 16-wave workgroup (RowsShape), at most BSR_RESID_WGS
    // (a data set far beyond the caches -- N = 1M: 977 row blocks -- is bound by what each CU can pull from HBM, ~25 GB/s:
    // 32 workgroups measured ~100 us whenever a proposal was flagged, 21 us on average with one batch in five flagging;
    // BSR_RESID_WGS_BIG workgroups there, whose empty launch costs a microsecond more)
    static const unsigned big = (unsigned)env_int("BSR_RESID_WGS_BIG", 128);
    grid.x = std::min<unsigned>(grid.x / 4, g.n_rb >= 512 ? std::max(big, BSR_RESID_WGS) : BSR_RESID_WGS);
    block.x = 4 * BSR_WG_WAVES * BSR_WAVE;
  }
  const int n_pg = (MODE == MODE_RESIDUAL) ? 1 : g.n_pg;
#ifdef BSR_TEST_VARIANTS
  if (a.feat_list) {   // (X staged in LDS per row block, BSR_NO_LDS=0: round 1's static grid, kept for the test build)
    const size_t lds = (size_t)(a.nF + 1) * g.rb_rows * sizeof(T);
    bsr_launch((k_rows<T, NQ, U, true, MODE>), grid, block, lds, st, a.Xt, a.y, a.ld, a.N, a.codes, a.feats,
                       a.lnp, a.desc, a.coef, a.P, g.rb_rows, g.pg, g.n_rb, n_pg, a.feat_list, a.nF, a.part,
                       (T*)a.spill, a.spill_slots, a.queue, a.queue_clear, a.fin);
  } else
#endif
  {
    bsr_launch((k_rows<T, NQ, U, false, MODE>), grid, block, 0, st, a.Xt, a.y, a.ld, a.N, a.codes, a.feats,
                       a.lnp, a.desc, a.coef, a.P, g.rb_rows, g.pg, g.n_rb, n_pg, a.feat_list, a.nF, a.part,
                       (T*)a.spill, a.spill_slots, a.queue, a.queue_clear, a.fin);
  }
}
template <typename T, int NQ, int MODE>
static void launch_rows_nq(hipStream_t st, const RowPassArgs<T>& a) {
  switch (a.rows_per_lane) {
#ifdef BSR_TEST_VARIANTS
    case 8: launch_rows_u<T, NQ, MODE, 8>(st, a); break;   // (BSR_P1_U: rows per lane and sweep, test build only)
    case 4: launch_rows_u<T, NQ, MODE, 4>(st, a); break;
#endif
    default: launch_rows_u<T, NQ, MODE, 2>(st, a); break;
  }
}
template <typename T>
void launch_rows(hipStream_t st, const RowPassArgs<T>& a, int nq, int residual) {
#define BSR_CASE(n)                                        \
  case n:                                                  \
    if (residual) launch_rows_nq<T, n, MODE_RESIDUAL>(st, a); \
    else launch_rows_nq<T, n, MODE_PROJECT>(st, a);        \
    break;
  switch (nq) {
    BSR_CASE(0) BSR_CASE(1) BSR_CASE(2) BSR_CASE(3) BSR_CASE(4) BSR_CASE(5) BSR_CASE(6) BSR_CASE(7) BSR_CASE(8)
  }
#undef BSR_CASE
}
void launch_solve(hipStream_t st, const PropDesc* desc, const ChainB* ck, int P, int n_rb, const double* part1,
                  int64_t N, PropCoef* coef, bsr_score* out, double rank_floor, int32_t* flagged, MhRes* mh,
                  int32_t* flagged_next) {
  bsr_launch(k_solve, dim3((P + BSR_SOLVE_WAVES - 1) / BSR_SOLVE_WAVES), dim3(BSR_SOLVE_WAVES * BSR_WAVE), 0, st, desc, ck, P, n_rb, part1, N, coef, out, rank_floor,
                     flagged, mh, flagged_next);
}
void launch_events(hipStream_t st, const MhRes* mh, const double* terms8, const int32_t* flags, const int32_t* span_off,
                   int n_spans, int K, bsr_event* events) {
  bsr_launch(k_events, dim3((n_spans + BSR_WAVE - 1) / BSR_WAVE), dim3(BSR_WAVE), 0, st, mh, terms8, flags,
                     span_off, n_spans, K, events);
}
void launch_finalize(hipStream_t st, const PropDesc* desc, const ChainB* ck, const PropCoef* coef, int P, int n_rb,
                     const double* part2, int64_t N, bsr_score* out, double rank_floor, int32_t* flagged, MhRes* mh,
                     int n_wg) {
  // a few proposals per batch at K=3 (one workgroup of four waves), a dozen or more at K=8, each ~10 us of one wave
  bsr_launch(k_finalize, dim3(n_wg), dim3(4 * BSR_WAVE), 0, st, desc, ck, coef, P, n_rb, part2, N, out,
                     rank_floor, flagged, mh);
}
template <typename T>
void launch_refresh_basis(hipStream_t st, const T* cur, T* Q, const T* y, int64_t ld, int64_t N, int K,
                          const double* col_maxabs, const uint32_t* col_flags, ChainB* cb) {
  hipLaunchKernelGGL((k_refresh_basis<T>), dim3(1), dim3(BSR_BLK), 0, st, cur, Q, y, ld, N, K, col_maxabs,
                     col_flags, cb);
}
template <typename T>
void launch_chain_fit(hipStream_t st, const T* cols, const T* y, int64_t ld, int64_t N, int K, int intercept,
                      ChainFitOut* out) {
  hipLaunchKernelGGL((k_chain_fit<T>), dim3(1), dim3(BSR_BLK), 0, st, cols, y, ld, N, K, intercept, out);
}
template <typename T>
void launch_transpose_in(hipStream_t st, const double* src_rowmajor, T* dst, int64_t N, int d, int64_t ld) {
  const int64_t nb = (N + 63) / 64;
  hipLaunchKernelGGL((k_transpose_in<T>), dim3((unsigned)nb), dim3(256), (size_t)64 * d * sizeof(double), st,
                     src_rowmajor, dst, N, d, ld);
}
template <typename T>
void launch_convert_out(hipStream_t st, const T* src, double* dst, int64_t n) {
  hipLaunchKernelGGL((k_convert_out<T>), dim3(1024), dim3(256), 0, st, src, dst, n);
}
template <typename T>
void launch_convert_in(hipStream_t st, const double* src, T* dst, int64_t n) {
  hipLaunchKernelGGL((k_convert_in<T>), dim3(1024), dim3(256), 0, st, src, dst, n);
}

#define BSR_INSTANTIATE(T)                                                                                          \
  template void launch_rows<T>(hipStream_t, const RowPassArgs<T>&, int, int);                                       \
  template void launch_refresh_basis<T>(hipStream_t, const T*, T*, const T*, int64_t, int64_t, int, const double*,  \
                                        const uint32_t*, ChainB*);                                                  \
  template void launch_chain_fit<T>(hipStream_t, const T*, const T*, int64_t, int64_t, int, int, ChainFitOut*);     \
  template void launch_transpose_in<T>(hipStream_t, const double*, T*, int64_t, int, int64_t);                      \
  template void launch_convert_out<T>(hipStream_t, const T*, double*, int64_t);                                     \
  template void launch_convert_in<T>(hipStream_t, const double*, T*, int64_t);
BSR_INSTANTIATE(double)
BSR_INSTANTIATE(float)
