// Tile-stationary, tape-streaming row pass for gfx950 (MI355X): the projection pass of the scoring path
// (allcal + the O(N) part of ylogLike / the rank gate, codes/funcs.py:175-220, 1147-1174, 1212-1226).
//
// One workgroup of 16 waves per CU.  A workgroup owns a row slice (a fixed share of the N rows) and one group of the
// batch's tapes.  It stages its rows of every column the batch needs -- the referenced X columns, y and the basis
// columns of the chains in the batch -- in LDS, chunk by chunk, and each of its waves runs "its" tapes (a static,
// cost-balanced schedule written by the host) over the staged rows: a lane owns 2 adjacent rows of every 128-row
// block, terminals are ds_read_b128, and the per-lane sums of a tape (projections on the basis, |s z|^2, s z.y,
// max|z|) stay in registers across blocks and chunks.  A tape is reduced over the lanes ONCE per slice, not once per
// (tape, row block) task as in k_rows: at N = 100k that is the difference between 4-8 k and 6 k x 5 wave reductions
// per launch, and there is no ticket queue, no per-task descriptor fetch and no exposed L2 latency per terminal.
//
// What is summed in which order depends only on the context (rows per lane, slice boundaries), never on the batch:
// a proposal's partial sums -- hence its score -- are bit-identical whatever else shares the launch.
#include <cstddef>
#include <type_traits>

#include "bsr_device.h"

namespace {

template <int KQ>
struct TapeAcc {
  double c[KQ > 0 ? KQ : 1];
  double a0, a1, amax;
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < KQ; ++i) c[i] = 0.0;
    a0 = a1 = amax = 0.0;
  }
};

// per-lane sums of one 128-row block of one tape; z: the lane's two candidate values, yv / qv: the lane's pair of y and
// of every basis column (read from LDS by the caller, early enough to be there when the tape has run)
template <typename T, int KQ, bool MASK>
__device__ __forceinline__ void accumulate_v(TapeAcc<KQ>& A, const T (&z)[BSR_TILE_U], const typename VecOf<T, 2>::type yv,
                                             const typename VecOf<T, 2>::type (&qv)[KQ > 0 ? KQ : 1], double s,
                                             int64_t row0, int64_t N) {
#pragma unroll
  for (int u = 0; u < BSR_TILE_U; ++u) {
    T zv = z[u];
    if (MASK) zv = (row0 + u < N) ? zv : (T)0;
    const double zd = (double)zv;
    const double zs = zd * s;
    A.amax = max_abs(A.amax, zd);
    A.a0 = fma(zs, zs, A.a0);
    A.a1 = fma(zs, (double)(u == 0 ? yv.x : yv.y), A.a1);
#pragma unroll
    for (int i = 0; i < KQ; ++i) A.c[i] = fma((double)(u == 0 ? qv[i].x : qv[i].y), zs, A.c[i]);
  }
}
template <typename T, int KQ, bool MASK>
__device__ __forceinline__ void accumulate(TapeAcc<KQ>& A, const T (&z)[BSR_TILE_U], const T* __restrict__ sy,
                                           const T* __restrict__ sq, int col_stride, double s, int64_t row0, int64_t N) {
  using V2 = typename VecOf<T, 2>::type;
  const V2 yv = *reinterpret_cast<const V2*>(sy);
  V2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
  for (int i = 0; i < KQ; ++i) qv[i] = *reinterpret_cast<const V2*>(sq + (size_t)i * col_stride);
  accumulate_v<T, KQ, MASK>(A, z, yv, qv, s, row0, N);
}

// Copies rows [c0*128, (c0+nb)*128) of every column of the launch into LDS.  Unit of work = 64 lanes x 16 B of one
// column; wave w takes units w, w+16, ... and keeps DEPTH loads in flight before it writes them to LDS.
// `mask`: the columns to stage (bit = LDS slot), ~0 = all of them.  A column's place in LDS does not depend on it.
template <typename T, int DEPTH>
__device__ __forceinline__ void stage_rows(T* sx, const T* const CONSTANT_AS* colsrc, int ncols, int chunk_rows, int c0,
                                           int nb, int wave, int lane, uint64_t mask = ~0ull) {
  constexpr int VEC = 16 / sizeof(T);
  using V4 = __attribute__((ext_vector_type(4))) float;
  constexpr int UPB = BSR_TILE_BLOCK / VEC / BSR_WAVE;      // units per column and block (f64: 1; f32: half a unit)
  const int upc = (UPB > 0) ? UPB * nb : (nb + 1) / 2;      // units per column
  const int nvec_col = nb * (BSR_TILE_BLOCK / VEC);         // 16-byte pieces per column
  if (mask != ~0ull) {   // a subset: walk its set bits instead of 0..ncols-1
    uint64_t m = mask;
    int ub = wave;
    while (ub >= upc && m != 0) { ub -= upc; m &= m - 1; }
    while (m != 0) {
      V4 r[DEPTH];
      int de[DEPTH];
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        de[j] = -1;
        if (m != 0) {
          const int col = __builtin_ctzll(m);
          const int piece = ub * BSR_WAVE + lane;
          if (piece < nvec_col) {
            const int e = piece * VEC;
            r[j] = *reinterpret_cast<const V4*>(colsrc[col] + (int64_t)c0 * BSR_TILE_BLOCK + e);
            de[j] = col * chunk_rows + e;
          }
          ub += BSR_TILE_WAVES;
          while (ub >= upc && m != 0) { ub -= upc; m &= m - 1; }
        }
      }
#pragma unroll
      for (int j = 0; j < DEPTH; ++j)
        if (de[j] >= 0) *reinterpret_cast<V4*>(sx + de[j]) = r[j];
    }
    return;
  }
  const int n_units = ncols * upc;
  int col = 0, ub = wave;                                   // unit = (col, ub): ub-th unit of column col
  while (ub >= upc && col < ncols) { ub -= upc; ++col; }
  for (int u0 = wave; u0 < n_units; u0 += DEPTH * BSR_TILE_WAVES) {
    V4 r[DEPTH];
    int de[DEPTH];                                          // LDS element offset of each piece, -1: none
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      de[j] = -1;
      if (col < ncols) {
        const int piece = ub * BSR_WAVE + lane;
        if (piece < nvec_col) {
          const int e = piece * VEC;                        // element offset inside the staged rows of the column
          r[j] = *reinterpret_cast<const V4*>(colsrc[col] + (int64_t)c0 * BSR_TILE_BLOCK + e);
          de[j] = col * chunk_rows + e;
        }
        ub += BSR_TILE_WAVES;
        while (ub >= upc && col < ncols) { ub -= upc; ++col; }
      }
    }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j)
      if (de[j] >= 0) *reinterpret_cast<V4*>(sx + de[j]) = r[j];
  }
}

// The same copy by LDS-DMA (fp64 columns): one instruction moves 64 lanes x 16 B of one column -- a 128-row block --
// from per-lane global addresses to 1 KiB of LDS, without registers.  Issued and left in flight; the caller's next
// workgroup barrier waits for it.
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <typename T>
__device__ __forceinline__ void dma_rows(T* buf, const T* const CONSTANT_AS* colsrc, int ncols, int chunk_rows, int c0,
                                         int nb, int wave, int lane) {
  static_assert(sizeof(T) == 8, "one 128-row block of a column per instruction");
  const int n_units = ncols * nb;
  int col = 0, blk = wave;
  while (blk >= nb && col < ncols) { blk -= nb; ++col; }
  for (int u = wave; u < n_units; u += BSR_TILE_WAVES) {
    const T* src = colsrc[col] + (int64_t)(c0 + blk) * BSR_TILE_BLOCK + 2 * lane;
    T* dst = buf + (size_t)col * chunk_rows + (size_t)blk * BSR_TILE_BLOCK;
    // Written as inline assembly on purpose: behind the builtin the compiler parks a vmcnt(0) in front of every later
    // LDS read (it cannot tell the two buffers apart), which turns the double buffer back into a single one.  The
    // caller waits for the copies itself (dma_wait) before the barrier that publishes them.
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(la) : "memory", "m0");
    blk += BSR_TILE_WAVES;
    while (blk >= nb && col < ncols) { blk -= nb; ++col; }
  }
}

// waits until at most `left` of the wave's copies are still in flight (copies complete in issue order)
__device__ __forceinline__ void dma_wait_left(int left) {
  switch (left) {
#define X(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
#undef X
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // 0, or more than the cases cover
  }
}

// Lane reduction of one tape's sums and the store of its (tape, slice) partial record.  Words 0..7: projections,
// 8: |s z|^2, 9: s z.y, 10: max|z|, 11: 0 (the inf / NaN census is taken from words 10 and 8 by k_solve).
template <int KQ>
__device__ __forceinline__ void store_partial(const TapeAcc<KQ>& A, double* o, int lane) {
  const int slot = ((lane >> 3) & 2) | (lane >> 5);   // rows 0, 1, 2, 3 of the wave hold values 0, 2, 1, 3 (in arithmetic:
                                                      // a table would be a memory load and a pointer kept in registers)
  double g0[8], lo0, hi0;
  if constexpr (KQ <= 6) {  // everything fits one group: c[0..KQ-1] at 0.., a0 at 6, a1 at 7
#pragma unroll
    for (int i = 0; i < 6; ++i) g0[i] = (i < KQ) ? A.c[i < KQ ? i : 0] : 0.0;
    g0[6] = A.a0;
    g0[7] = A.a1;
    wave_sum8_swz(g0, lo0, hi0);
    const double amax = wave_max_swz_hi(A.amax);
    if ((lane & 15) == 0) {
      o[slot] = lo0;                                   // values 0..3
      const int hslot = 4 + slot;                      // values 4..7 -> words 4, 5 (projections) and 8, 9 (a0, a1)
      o[hslot < 6 ? hslot : hslot + 2] = hi0;
    }
    if (lane == 0) {
      o[6] = 0.0;
      o[7] = 0.0;
    }
    if (lane == 63) {
      o[10] = amax;
      o[11] = 0.0;
    }
  } else {
    double g1[8], lo1, hi1;
#pragma unroll
    for (int i = 0; i < 8; ++i) g0[i] = (i < KQ) ? A.c[i < KQ ? i : 0] : 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) g1[i] = 0.0;
    g1[0] = A.a0;
    g1[1] = A.a1;
    wave_sum8_swz(g0, lo0, hi0);
    wave_sum8_swz(g1, lo1, hi1);
    const double amax = wave_max_swz_hi(A.amax);
    if ((lane & 15) == 0) {
      o[slot] = lo0;
      o[4 + slot] = hi0;
      if (slot < 2) o[8 + slot] = lo1;                 // values 0 (a0, row 0) and 1 (a1, row 2 -> slot 1)
    }
    if (lane == 63) {
      o[10] = amax;
      o[11] = 0.0;
    }
  }
}

// Column reads straight from global memory (L2) through the launch's column-pointer table, by LDS slot id: the loader of
// the leftover units, whose rows are in no workgroup's LDS.
template <typename T, int U>
struct PtrCols {
  const T* const CONSTANT_AS* colsrc;
  int64_t r0;  // absolute row of the lane's pair
  __device__ __forceinline__ void load(int slot, T (&v)[U]) const {
    const T* col = colsrc[slot] + r0;
#pragma unroll
    for (int j = 0; j < U / 2; ++j) {
      v[2 * j] = col[j * 128];
      v[2 * j + 1] = col[j * 128 + 1];
    }
  }
};

// The blocks behind the last slice (n_blocks is rarely a multiple of the slice count; at N = 100k: 14 of 782) as
// (tape, block) units: tapes in cost order, every unit one single-block pass with its own partial record (index
// n_slices * n_sub + block).  Unit u belongs to workgroup u mod n_wg (3 or 4 units each at C2 instead of a seventh
// block for 28 workgroups); inside the workgroup the waves take them through the same LDS counter that hands out the
// tapes, so they go to whoever runs dry first.  (A global ticket counter was tried first: 4 096 waves on one address
// serialise at the memory side, 50 us.)  `li` is the wave's first leftover item, `counter` the LDS counter, `base`
// its value at the first leftover item.
template <typename T, int KQ>
__device__ __forceinline__ void leftover_units(const TileArgs<T>& a, const TileGeom& g, int lane, int li, int* counter,
                                               int base) {
  constexpr int U = BSR_TILE_U;
  constexpr int S = BSR_REG_STACK;
  using V2 = typename VecOf<T, 2>::type;
  const PropDesc CONSTANT_AS* dsc = as_const(a.desc);
  const T* const CONSTANT_AS* colsrc = (const T* const CONSTANT_AS*)a.colsrc;
  const int n_units = a.P * g.n_left, n_wg = (int)gridDim.x;
  for (;;) {
    const int tk = li * n_wg + (int)blockIdx.x;
    if (tk >= n_units) break;
    int nx = 0;
    if (lane == 0) nx = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int ti = tk / g.n_left, bi = tk - ti * g.n_left;
    const int p = dsc[ti].order;   // the ti-th most expensive tape
    const int blk = g.n_slices * g.bps + bi;
    const int64_t row0 = (int64_t)blk * BSR_TILE_BLOCK + 2 * lane;
    const uint64_t* pc = a.codes + dsc[p].code_off;
    const uint64_t* pf = a.feats + dsc[p].feat_off;
    const double* pl = a.lnp + 2 * (size_t)dsc[p].ln_off;
    const int qslot = dsc[p].qslot;
    const V2 yv = *reinterpret_cast<const V2*>(colsrc[g.y_slot] + row0);
    V2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
    for (int i = 0; i < KQ; ++i) qv[i] = *reinterpret_cast<const V2*>(colsrc[qslot + i] + row0);
    T z[U];
    PtrCols<T, U> ldr{colsrc, row0};
    run_tape<T, U, S>(pc, pf, pl, dsc[p].n_nodes, ldr, z, (T*)nullptr, lane);
    TapeAcc<KQ> A;
    A.clear();
    if ((int64_t)(blk + 1) * BSR_TILE_BLOCK <= a.N) accumulate_v<T, KQ, false>(A, z, yv, qv, dsc[p].s, row0, a.N);
    else accumulate_v<T, KQ, true>(A, z, yv, qv, dsc[p].s, row0, a.N);
    store_partial<KQ>(A, a.part + ((size_t)p * g.n_part + g.n_slices + bi) * BSR_P1_WORDS, lane);
    li = __builtin_amdgcn_readfirstlane(nx) - base;
  }
}

// Single-chunk variant: the workgroup's whole slice fits in LDS.  Staged once; then the waves pull tapes from the
// group's list (heaviest first) through an LDS counter, and a wave runs its tape over all blocks of the slice with one
// set of accumulators.  Which wave runs a tape does not matter to the sums (one wave, blocks in order).
//
// A chain tape (bsr_device.h: chain_eval) is run a pass of up to NBIG blocks at a time, the whole pass in registers:
// one decode of the tape per pass, operators in place.  Any other tape goes through the stack machine two blocks at a
// time.  Both produce the same values per row and add them up in the same order (per lane: blocks in order, the
// lane's two rows of a block in order), so a proposal's sums do not depend on which route its tape takes.
template <typename T, int KQ>
__global__ __launch_bounds__(BSR_TILE_WAVES* BSR_WAVE) void k_tile1(TileArgs<T> a) {
  constexpr int U = BSR_TILE_U;
  constexpr int S = BSR_REG_STACK;
  // blocks per pass of a chain tape: what the register file holds next to the accumulators (f64: K <= 4 eight blocks
  // = 32 VGPRs of values, K >= 5 four)
  constexpr int NBIG = (sizeof(T) == 4 || KQ <= 4) ? 8 : 4;
  using V2 = typename VecOf<T, 2>::type;
  extern __shared__ __align__(16) unsigned char smem[];
  T* sx = reinterpret_cast<T*>(smem);  // [ncols][chunk_rows]
  __shared__ int s_next;
  const TileGeom g = a.g;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tg = blockIdx.x / g.n_slices, slice = blockIdx.x - tg * g.n_slices;
  // every slice holds bps blocks; the n_left blocks behind the last slice are handed out one (tape, block) at a time
  // to whichever wave runs dry first (leftover_units): no workgroup carries a block more than the others
  const int b0 = slice * g.bps;
  const int nb = g.bps;
  const int chunk_rows = g.chunk_blocks * BSR_TILE_BLOCK;
  const PropDesc CONSTANT_AS* dsc = as_const(a.desc);
  const int32_t CONSTANT_AS* list = as_const(a.sched) + (size_t)tg * g.per_group;
  const T* const CONSTANT_AS* colsrc = (const T* const CONSTANT_AS*)a.colsrc;
  const T* sy = sx + (size_t)g.y_slot * chunk_rows;
  const int n_full = (int)(a.N / BSR_TILE_BLOCK);   // blocks that lie below row N whole
  unsigned long long* stamp = a.stamps ? a.stamps + ((size_t)blockIdx.x * BSR_TILE_WAVES + wave) * BSR_TILE_STAMP_WORDS : nullptr;
#define TSTAMP(i) do { if (stamp && lane == 0) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
  TSTAMP(0);
  if (stamp && lane == 0) stamp[7] = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) s_next = BSR_TILE_WAVES;
  tables_to_lds();
  stage_rows<T, 8>(sx, colsrc, g.ncols, chunk_rows, b0, nb, wave, lane, a.grp_mask[tg & 7]);
  __syncthreads();
  TSTAMP(1);
  TSTAMP(2);
  // unit of work: a tape of the group's list over the whole slice, heaviest tape first.  (Finer units -- every tape, or
  // only the heavy ones, in two halves -- were measured twice: the lane reduction and the decode each extra unit pays
  // cost what the better balance gains.)
  const int n_items = g.per_group;
  int idx = wave;   // the first list entries go to the waves in order
  while (idx < n_items) {
    const int p = list[idx];
    if (p < 0) {   // padding behind the group's last tape: take the next item (the leftover units follow the list)
      int nx = 0;
      if (lane == 0) nx = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      idx = __builtin_amdgcn_readfirstlane(nx);
      continue;
    }
    const int sb0 = 0, sb1 = nb;
    // the next unit is requested now; its round trip hides under this one
    int nxt = 0;
    if (lane == 0) nxt = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint64_t* pc = a.codes + dsc[p].code_off;
    const uint64_t* pf = a.feats + dsc[p].feat_off;
    const double* pl = a.lnp + 2 * (size_t)dsc[p].ln_off;
    const int n_nodes = dsc[p].n_nodes;
    const double s = dsc[p].s;
    const bool chain = dsc[p].chain != 0;
    const T* sq = sx + (size_t)dsc[p].qslot * chunk_rows;
    const TapeHead hd = load_tape_head(pc, pf, pl);
    TapeAcc<KQ> A;
    A.clear();
    // one 128-row block: the lane's pair of rows, sums in row order
    auto add_block = [&](const T (&zz)[U], int off, int b) {
      const int64_t row0 = (int64_t)(b0 + b) * BSR_TILE_BLOCK + 2 * lane;
      const V2 yv = *reinterpret_cast<const V2*>(sy + off);
      V2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
      for (int i = 0; i < KQ; ++i) qv[i] = *reinterpret_cast<const V2*>(sq + (size_t)i * chunk_rows + off);
      if (b0 + b < n_full) accumulate_v<T, KQ, false>(A, zz, yv, qv, s, row0, a.N);
      else accumulate_v<T, KQ, true>(A, zz, yv, qv, s, row0, a.N);
    };
    int b = sb0;
    if (chain) {
      // a pass of NB blocks: the tape once over 2 NB values per lane, then the blocks' sums in order
      auto pass = [&](auto nb_tag, auto full_tag, int pn) {
        constexpr int NB = decltype(nb_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value;
        const int off = b * BSR_TILE_BLOCK + 2 * lane;  // the lane's pair inside the slice (block j of the pass: 128 j rows on)
        T z[2 * NB];
        chain_eval<T, NB, FULL>(hd, pc, pf, pl, n_nodes, sx, chunk_rows, off, pn, z);
        const T* yp = sy + off;
        const T* qp = sq + off;
        const bool whole = b0 + b + pn <= n_full;   // no block of the pass reaches beyond row N
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          if (FULL || jb < pn) {
            const T zz[U] = {z[2 * jb], z[2 * jb + 1]};
            const V2 yv = *reinterpret_cast<const V2*>(yp + jb * BSR_TILE_BLOCK);
            V2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
            for (int i = 0; i < KQ; ++i) qv[i] = *reinterpret_cast<const V2*>(qp + (size_t)i * chunk_rows + jb * BSR_TILE_BLOCK);
            const int64_t row0 = (int64_t)(b0 + b + jb) * BSR_TILE_BLOCK + 2 * lane;
            if (whole || b0 + b + jb < n_full) accumulate_v<T, KQ, false>(A, zz, yv, qv, s, row0, a.N);
            else accumulate_v<T, KQ, true>(A, zz, yv, qv, s, row0, a.N);
          }
        }
        b += pn;
      };
      using std::integral_constant;
      while (NBIG > 4 && sb1 - b >= NBIG) pass(integral_constant<int, NBIG>{}, integral_constant<bool, true>{}, NBIG);
      while (sb1 - b >= 4) pass(integral_constant<int, 4>{}, integral_constant<bool, true>{}, 4);
      if (b < sb1) pass(integral_constant<int, 4>{}, integral_constant<bool, false>{}, sb1 - b);
    } else {
      // Two blocks per interpreter pass (4 rows per lane): the scalar decode of a node is paid once per 256 rows.  The
      // per-lane sums still grow block by block in row order, so the result does not depend on the pairing.
#pragma unroll 1
      for (; b + 1 < sb1; b += 2) {
        const int off = b * BSR_TILE_BLOCK + 2 * lane;  // the lane's pair inside the slice (second pair 128 rows on)
        T z4[2 * U];
        LdsCols<T, 2 * U> ldr{sx, chunk_rows, off};
        // four values per out-of-line call, except where that costs the last registers (K = 6, 8 would spill two)
        run_tape_head<T, 2 * U, S, LdsCols<T, 2 * U>, (sizeof(T) == 4 || (KQ != 6 && KQ != 8))>(hd, pc, pf, pl, n_nodes, ldr, z4, (T*)nullptr, lane);
        const T za[U] = {z4[0], z4[1]}, zb[U] = {z4[2], z4[3]};
        add_block(za, off, b);
        add_block(zb, off + BSR_TILE_BLOCK, b + 1);
      }
      if (b < sb1) {
        const int off = b * BSR_TILE_BLOCK + 2 * lane;
        T z[U];
        LdsCols<T, U> ldr{sx, chunk_rows, off};
        run_tape_head<T, U, S>(hd, pc, pf, pl, n_nodes, ldr, z, (T*)nullptr, lane);
        add_block(z, off, b);
      }
    }
    store_partial<KQ>(A, a.part + ((size_t)p * g.n_part + slice) * BSR_P1_WORDS, lane);
    idx = __builtin_amdgcn_readfirstlane(nxt);
  }
  TSTAMP(3);
  // the tape list is empty: idx - n_items is this wave's first item of the workgroup's leftover units
  if (g.n_left > 0 && idx >= n_items) leftover_units<T, KQ>(a, g, lane, idx - n_items, &s_next, n_items);
  TSTAMP(4);
  if (stamp && lane == 0) stamp[6] = __builtin_amdgcn_s_memrealtime();
#undef TSTAMP
}

template <typename T, int KQ, int QMAX>
__global__ __launch_bounds__(BSR_TILE_WAVES* BSR_WAVE) void k_tile(TileArgs<T> a) {
  constexpr int U = BSR_TILE_U;
  constexpr int S = BSR_REG_STACK;
  extern __shared__ __align__(16) unsigned char smem[];
  T* sx = reinterpret_cast<T*>(smem);  // [ncols][chunk_rows]
  const TileGeom g = a.g;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tg = blockIdx.x / g.n_slices, slice = blockIdx.x - tg * g.n_slices;
  const int b0 = slice * g.bps;   // slices of bps blocks; the blocks behind the last one: leftover_units
  const int b1 = b0 + g.bps;
  const int chunk_rows = g.chunk_blocks * BSR_TILE_BLOCK;
  const PropDesc CONSTANT_AS* dsc = as_const(a.desc);
  const int32_t CONSTANT_AS* sched = as_const(a.sched);
  const T* const CONSTANT_AS* colsrc = (const T* const CONSTANT_AS*)a.colsrc;
  // diagnostics: shader-clock samples per wave (0 start, 1 first chunk staged, 2 first chunk computed, 3 all chunks
  // computed, 4 reductions stored; 7 and 6: the constant-rate 100 MHz clock at the start and at the end -- the shader
  // clock counters of different XCDs are not aligned, only differences inside one wave mean anything)
  unsigned long long* stamp = a.stamps ? a.stamps + ((size_t)blockIdx.x * BSR_TILE_WAVES + wave) * BSR_TILE_STAMP_WORDS : nullptr;
#define TSTAMP(i) do { if (stamp && lane == 0) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
  TSTAMP(0);
  if (stamp && lane == 0) stamp[7] = __builtin_amdgcn_s_memrealtime();

  __shared__ int s_left;   // hands the workgroup's leftover units to its waves
  if (threadIdx.x == 0) s_left = 0;
  tables_to_lds();  // visible after the first barrier below
  // Two LDS buffers of chunk_rows rows per column.  While the waves run their tapes over chunk c, the rows of chunk
  // c+1 travel HBM -> LDS on their own (LDS-DMA: no registers, nothing for the waves to do); the barrier that ends the
  // chunk also waits for them (its fence drains vmcnt).  f32 columns keep the register-staged single buffer.
  constexpr bool DMA = sizeof(T) == 8;
  const int buf_elems = DMA ? g.ncols * chunk_rows : 0;
  for (int pass = 0; pass < g.n_pass; ++pass) {
    const int32_t CONSTANT_AS* my = sched + (((size_t)tg * g.n_pass + pass) * BSR_TILE_WAVES + wave) * QMAX;
    TapeAcc<KQ> A[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; ++q) A[q].clear();

    if constexpr (DMA) {
      if (pass != 0) __syncthreads();  // everyone is done with the last chunk of the pass before
      dma_rows<T>(sx, colsrc, g.ncols, chunk_rows, b0, min(g.chunk_blocks, b1 - b0), wave, lane);
      dma_wait();
      __syncthreads();
      if (pass == 0) TSTAMP(1);
    }
    int ci = 0;
    for (int c0 = b0; c0 < b1; c0 += g.chunk_blocks, ++ci) {
      const int nb = min(g.chunk_blocks, b1 - c0);
      const T* cur = sx + (size_t)(ci & 1) * buf_elems;
      if constexpr (DMA) {
        const int n0 = c0 + g.chunk_blocks;
        if (n0 < b1)
          dma_rows<T>(sx + (size_t)((ci + 1) & 1) * buf_elems, colsrc, g.ncols, chunk_rows, n0, min(g.chunk_blocks, b1 - n0),
                      wave, lane);
      } else {
        if (c0 != b0 || pass != 0) __syncthreads();  // everyone is done with the rows staged before
        stage_rows<T, 4>(sx, colsrc, g.ncols, chunk_rows, c0, nb, wave, lane);
        __syncthreads();
        if (c0 == b0 && pass == 0) TSTAMP(1);
      }
      const T* sy = cur + (size_t)g.y_slot * chunk_rows;
#pragma unroll 1
      for (int q = 0; q < QMAX; ++q) {
        const int p = my[q];
        if (p < 0) continue;
        const uint64_t* pc = a.codes + dsc[p].code_off;
        const uint64_t* pf = a.feats + dsc[p].feat_off;
        const double* pl = a.lnp + 2 * (size_t)dsc[p].ln_off;
        const int n_nodes = dsc[p].n_nodes;
        const double s = dsc[p].s;
        const T* sq = cur + (size_t)dsc[p].qslot * chunk_rows;
#pragma unroll 1
        for (int b = 0; b < nb; ++b) {
          const int off = b * BSR_TILE_BLOCK + 2 * lane;  // the lane's pair inside the chunk
          const int64_t row0 = (int64_t)(c0 + b) * BSR_TILE_BLOCK + 2 * lane;
          T z[U];
          LdsCols<T, U> ldr{cur, chunk_rows, off};
          run_tape<T, U, S>(pc, pf, pl, n_nodes, ldr, z, (T*)nullptr, lane);
          const bool full = (int64_t)(c0 + b + 1) * BSR_TILE_BLOCK <= a.N;  // wave-uniform
#define BSR_ACC_CASE(qq)                                                                              \
  case qq:                                                                                            \
    if constexpr (qq < QMAX) {                                                                        \
      if (full) accumulate<T, KQ, false>(A[qq], z, sy + off, sq + off, chunk_rows, s, row0, a.N);     \
      else accumulate<T, KQ, true>(A[qq], z, sy + off, sq + off, chunk_rows, s, row0, a.N);           \
    }                                                                                                 \
    break;
          switch (q) { BSR_ACC_CASE(0) BSR_ACC_CASE(1) BSR_ACC_CASE(2) BSR_ACC_CASE(3) }
#undef BSR_ACC_CASE
        }
      }
      if constexpr (DMA) {  // chunk done by every wave, and the next one has landed
        dma_wait();
        __syncthreads();
      }
      if (c0 == b0 && pass == 0) TSTAMP(2);
    }
    if (pass == g.n_pass - 1) TSTAMP(3);
    // one lane reduction per (tape, slice); every lane stores the same totals (no lane-divergent branch)
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
      const int p = my[q];
      if (p < 0) continue;
      store_partial<KQ>(A[q], a.part + ((size_t)p * g.n_part + slice) * BSR_P1_WORDS, lane);
    }
  }
  if (g.n_left > 0) {
    int li = 0;
    if (lane == 0) li = __hip_atomic_fetch_add(&s_left, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    leftover_units<T, KQ>(a, g, lane, __builtin_amdgcn_readfirstlane(li), &s_left, 0);
  }
  TSTAMP(4);
  if (stamp && lane == 0) stamp[6] = __builtin_amdgcn_s_memrealtime();
#undef TSTAMP
}

template <typename T, int KQ>
void launch_kq(hipStream_t st, const TileArgs<T>& a) {
  const TileGeom& g = a.g;
  const dim3 grid((unsigned)(g.T * g.n_slices)), block(BSR_TILE_WAVES * BSR_WAVE);
  // the chunked fp64 variant keeps two buffers (LDS-DMA double buffering)
  const size_t lds = (size_t)g.ncols * g.chunk_blocks * BSR_TILE_BLOCK * sizeof(T) * ((g.per_group == 0 && sizeof(T) == 8) ? 2 : 1);
  if (g.per_group > 0) {
    static bool attr0 = false;
    if (!attr0) {
      (void)hipFuncSetAttribute((const void*)k_tile1<T, KQ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(tile_lds_bytes_max() - 1024));
      attr0 = true;
    }
    hipLaunchKernelGGL((k_tile1<T, KQ>), grid, block, lds, st, a);
  } else if (g.qmax == 1) {
    static bool attr1 = false;
    if (!attr1) {
      (void)hipFuncSetAttribute((const void*)k_tile<T, KQ, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(tile_lds_bytes_max() - 1024));
      attr1 = true;
    }
    hipLaunchKernelGGL((k_tile<T, KQ, 1>), grid, block, lds, st, a);
  } else {
    constexpr int QB = (KQ <= 3) ? 4 : ((KQ <= 5) ? 3 : 2);
    static bool attrb = false;
    if (!attrb) {
      (void)hipFuncSetAttribute((const void*)k_tile<T, KQ, QB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(tile_lds_bytes_max() - 1024));
      attrb = true;
    }
    hipLaunchKernelGGL((k_tile<T, KQ, QB>), grid, block, lds, st, a);
  }
}

}  // namespace

size_t tile_lds_bytes_max() { return 160 * 1024 - BSR_TAB_DOUBLES * sizeof(double); }  // the math tables are static LDS

template <typename T>
void launch_tile(hipStream_t st, const TileArgs<T>& a) {
  switch (a.K) {
    case 1: launch_kq<T, 1>(st, a); break;
    case 2: launch_kq<T, 2>(st, a); break;
    case 3: launch_kq<T, 3>(st, a); break;
    case 4: launch_kq<T, 4>(st, a); break;
    case 5: launch_kq<T, 5>(st, a); break;
    case 6: launch_kq<T, 6>(st, a); break;
    case 7: launch_kq<T, 7>(st, a); break;
    default: launch_kq<T, 8>(st, a); break;
  }
}
template void launch_tile<double>(hipStream_t, const TileArgs<double>&);
template void launch_tile<float>(hipStream_t, const TileArgs<float>&);
