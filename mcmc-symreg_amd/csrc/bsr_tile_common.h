// Building blocks shared by the tile row passes (bsr_tile.hip: k_tile1 / k_tile, bsr_stream.hip: k_stream): per-lane sums of a
// tape, the LDS staging and LDS-DMA helpers, the lane reductions and the store of the (tape, slice) partial records.
#pragma once
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
__device__ __forceinline__ int dma_rows(T* buf, const T* const CONSTANT_AS* colsrc, int ncols, int chunk_rows, int c0,
                                         int nb, int wave, int lane, uint64_t mask = ~0ull) {
  static_assert(sizeof(T) == 8, "one 128-row block of a column per instruction");
  // unit = (column, block), column-major over the columns to stage (`mask`: bit = LDS slot, all ones = every column)
  const int n_sel = (mask == ~0ull) ? ncols : __builtin_popcountll(mask);
  const int n_units = n_sel * nb;
  int issued = 0;
  for (int u = wave; u < n_units; u += BSR_TILE_WAVES) {
    ++issued;
    const int ci = u / nb, blk = u - ci * nb;
    int col = ci;
    if (mask != ~0ull) {
      uint64_t m = mask;
      for (int k = 0; k < ci; ++k) m &= m - 1;
      col = __builtin_ctzll(m);
    }
    const T* src = colsrc[col] + (int64_t)(c0 + blk) * BSR_TILE_BLOCK + 2 * lane;
    T* dst = buf + (size_t)col * chunk_rows + (size_t)blk * BSR_TILE_BLOCK;
    // Written as inline assembly on purpose: behind the builtin the compiler parks a vmcnt(0) in front of every later
    // LDS read (it cannot tell the two buffers apart), which turns the double buffer back into a single one.  The
    // caller waits for the copies itself (dma_wait) before the barrier that publishes them.
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(la) : "memory", "m0");
  }
  return issued;
}

// waits until at most `left` of the wave's copies are still in flight (copies complete in issue order)
__device__ __forceinline__ void dma_wait_left(int left) {
  switch (left) {
#define X(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
    X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40)
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
  if constexpr (KQ == 3) {
    // five sums: (c0 c1) and (c2 a0) through the swap network, a1 on its own (its partner lanes come by ds_swizzle);
    // every total is combined in the order of the general network below -- halves, rows 16 apart, then lane ^ 1, 2,
    // 4, 8 -- so the bits are the same, for a third fewer vector instructions than eight padded slots
    double v0 = A.c[0], v1 = A.c[1], v2 = A.c[2], v3 = A.a0, v4 = A.a1, v5 = A.a1;
    swap32(v0, v1);
    swap32(v2, v3);
    swap32(v4, v5);
    double w0 = v0 + v1, w1 = v2 + v3;
    const double w2 = v4 + v5;                       // every lane: a1 over both halves
    swap16(w0, w1);
    lo0 = row_sum16_swz(w0 + w1);                     // rows 0, 1, 2, 3: c0, c2, c1, a0
    hi0 = row_sum16_swz(w2 + swz_xor_f64<16>(w2));    // every row: a1
    const double amax = wave_max_swz_hi(A.amax);
    if ((lane & 15) == 0) o[slot < 3 ? slot : 8] = lo0;
    if (lane == 0) {
      o[3] = 0.0; o[4] = 0.0; o[5] = 0.0; o[6] = 0.0; o[7] = 0.0;
      o[9] = hi0;
    }
    if (lane == 63) {
      o[10] = amax;
      o[11] = 0.0;
    }
  } else if constexpr (KQ <= 6) {  // everything fits one group: c[0..KQ-1] at 0.., a0 at 6, a1 at 7
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

// Lane reduction of the sums of the QMAX tapes a wave ran, all at once, and the store of their (tape, slice) records.
// The NV = KQ + 2 sums of every tape go through ONE swap network, densely packed: v_permlane32_swap + add halves the
// lanes of two quantities at a time, v_permlane16_swap + add again, and the last four steps inside a 16-lane row run
// on the LDS crossbar (ds_swizzle) -- for K = 3, four tapes: 20 sums in five registers' worth of in-row steps,
// ~20 vector instructions per tape where one tape at a time took ~45.  max|z| likewise with maxima.  The totals land
// in a per-wave LDS scratch, from where lane (q, word) writes word `word` of tape q's record.
// Every sum is combined in the same order whatever else shares the network (halves, 16-lane rows, then lane ^ 1, 2, 4,
// 8): a tape's totals do not depend on its companions.
template <int KQ, int QMAX>
__device__ __forceinline__ void reduce_store(const TapeAcc<KQ> (&A)[QMAX], const TapeRec* my, double* part,
                                             int n_part, int rec, double* scratch, int lane) {
  constexpr int NV = KQ + 2;
  constexpr int NT = QMAX * NV;
  constexpr int NTP = (NT + 3) / 4 * 4;
  double v[NTP];
#pragma unroll
  for (int q = 0; q < QMAX; ++q) {
#pragma unroll
    for (int i = 0; i < KQ; ++i) v[q * NV + i] = A[q].c[i];
    v[q * NV + KQ] = A[q].a0;
    v[q * NV + KQ + 1] = A[q].a1;
  }
#pragma unroll
  for (int i = NT; i < NTP; ++i) v[i] = 0.0;
  const int row = lane >> 4;
  const int rmap = ((row & 1) << 1) | (row >> 1);   // rows 0, 1, 2, 3 of a register end with values 0, 2, 1, 3 of its four
  const bool writer = (lane & 15) == 0;
#pragma unroll
  for (int i = 0; i < NTP / 4; ++i) {
    double w0, w1;
    swap32(v[4 * i], v[4 * i + 1]);
    w0 = v[4 * i] + v[4 * i + 1];
    swap32(v[4 * i + 2], v[4 * i + 3]);
    w1 = v[4 * i + 2] + v[4 * i + 3];
    swap16(w0, w1);
    const double x = row_sum16_swz(w0 + w1);
    if (writer) scratch[4 * i + rmap] = x;
  }
  // max|z|: the same network with maxima (inputs are non-negative)
  if constexpr (QMAX == 4) {
    double m0 = A[0].amax, m1 = A[1].amax, m2 = A[2].amax, m3 = A[3].amax;
    swap32(m0, m1);
    m0 = vmax_raw(m0, m1);
    swap32(m2, m3);
    m2 = vmax_raw(m2, m3);
    swap16(m0, m2);
    double m = vmax_raw(m0, m2);
    m = vmax_raw(m, swz_xor_f64<1>(m));
    m = vmax_raw(m, swz_xor_f64<2>(m));
    m = vmax_raw(m, swz_xor_f64<4>(m));
    m = vmax_raw(m, swz_xor_f64<8>(m));
    if (writer) scratch[NTP + rmap] = m;
  } else {
    static_assert(QMAX == 2, "two or four tapes per wave");
    double m0 = A[0].amax, m1 = A[1].amax;
    swap32(m0, m1);
    double m = vmax_raw(m0, m1);     // lanes 0..31: tape 0, lanes 32..63: tape 1
    m = vmax_raw(m, swz_xor_f64<1>(m));
    m = vmax_raw(m, swz_xor_f64<2>(m));
    m = vmax_raw(m, swz_xor_f64<4>(m));
    m = vmax_raw(m, swz_xor_f64<8>(m));
    m = vmax_raw(m, swz_xor_f64<16>(m));
    if ((lane & 31) == 0) scratch[NTP + (lane >> 5)] = m;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed (waves do not share scratch)
  // lane (q, w): word w of tape q's record -- 0..7 projections, 8 |s z|^2, 9 s z.y, 10 max|z|, 11 zero
  if (lane < QMAX * BSR_P1_WORDS) {
    const int q = lane / BSR_P1_WORDS, w = lane - q * BSR_P1_WORDS;
    const int p = my[q].p;
    const int idx = (w < KQ) ? q * NV + w : (w == 8) ? q * NV + KQ : (w == 9) ? q * NV + KQ + 1 : NTP + q;
    const bool has = (w < KQ) || (w >= 8 && w <= 10);
    const double val = has ? scratch[idx] : 0.0;
    if (p >= 0) part[((size_t)p * n_part + rec) * BSR_P1_WORDS + w] = val;
  }
  __builtin_amdgcn_wave_barrier();   // the scratch is reused by the wave's next pass
}

// a tape group's column-pointer table: from the kernel-argument block where it fits there.  A kernel names its own
// argument block through the builtin (its by-value parameter has no address worth taking); a callee that was handed the
// block as a pointer into the constant address space (AT qualified so) takes the table's address from that.
template <typename T, typename AT = TileArgs<T>>
__device__ __forceinline__ const T* const CONSTANT_AS* group_cols(const AT& a, int grp) {
  const char CONSTANT_AS* base;
  if constexpr (std::is_same<AT, TileArgs<T>>::value) base = (const char CONSTANT_AS*)__builtin_amdgcn_kernarg_segment_ptr();
  else base = (const char CONSTANT_AS*)&a;
  return a.cols_in_args ? (const T* const CONSTANT_AS*)(base + offsetof(TileArgs<T>, cols)) + (size_t)grp * a.arg_stride
                        : (const T* const CONSTANT_AS*)a.colsrc + (size_t)grp * a.cols_stride;
}

// The blocks behind the last slice (n_blocks is rarely a multiple of the slice count; at N = 100k: 14 of 782) as
// (tape, block) units: tapes in cost order, every unit one single-block pass with its own partial record (index
// n_slices + block), dealt to the waves of the launch in order: unit u goes to wave u mod (workgroups x 16) -- at C2
// 896 units over 3 072 waves instead of a fifth block for some workgroups.  Their rows are in no workgroup's LDS:
// columns are read through the launch's column-pointer table.
// (AT: TileArgs<T>, or the same in the constant address space for a caller that reads its arguments through a pointer)
template <typename T, int KQ, typename AT = TileArgs<T>, typename GT = TileGeom>
__device__ __forceinline__ void leftover_unit(const AT& a, const GT& g, int lane, int tk) {
  constexpr int U = BSR_TILE_U;
  constexpr int S = BSR_REG_STACK;
  using V2 = typename VecOf<T, 2>::type;
  // the tapes in cost order: index of the tape's record in the schedule (the index list sits behind the records)
  const int32_t CONSTANT_AS* left_idx = as_const(reinterpret_cast<const int32_t*>(a.sched + (size_t)g.T * g.n_pass * BSR_TILE_WAVES * g.qmax));
  const int ti = tk / g.n_left, bi = tk - ti * g.n_left;
  const TapeRec CONSTANT_AS* rec = as_const(a.sched + left_idx[ti]);   // the ti-th most expensive tape
  const int p = rec->p;
  const int blk = g.n_slices * g.bps + g.n_long + bi;
  const int64_t row0 = (int64_t)blk * BSR_TILE_BLOCK + 2 * lane;
  const uint64_t* pc = a.codes + rec->code_off;
  const uint64_t* pf = a.feats + rec->feat_off;
  const double* pl = a.lnp + 2 * (size_t)rec->ln_off;
  const int qslot = rec->qslot, grp = rec->grp;
  const T* const CONSTANT_AS* colsrc = group_cols<T, AT>(a, grp);   // the tape's group
  const V2 yv = *reinterpret_cast<const V2*>(colsrc[a.grp_nF[grp & 7]] + row0);
  V2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
  for (int i = 0; i < KQ; ++i) qv[i] = *reinterpret_cast<const V2*>(colsrc[qslot + i] + row0);
  TapeHead hd;
  hd.code0 = rec->code0; hd.code1 = rec->code1; hd.f0 = rec->f0; hd.f1 = rec->f1;
  hd.la = rec->ln[0]; hd.lb = rec->ln[1];
  hd.ln_near = (const double*)rec->ln;
  hd.n_ln = rec->n_ln;
  hd.n_term = rec->n_term;
  const double s = rec->s;
  T z[U];
  PtrCols<T, U> ldr{colsrc, row0};
  run_tape_head<T, U, S, PtrCols<T, U>, false>(hd, pc, pf, pl, rec->n_nodes, ldr, z, (T*)nullptr, lane);
  TapeAcc<KQ> A;
  A.clear();
  if ((int64_t)(blk + 1) * BSR_TILE_BLOCK <= a.N) accumulate_v<T, KQ, false>(A, z, yv, qv, s, row0, a.N);
  else accumulate_v<T, KQ, true>(A, z, yv, qv, s, row0, a.N);
  store_partial<KQ>(A, a.part + ((size_t)p * g.n_part + g.n_slices + bi) * BSR_P1_WORDS, lane);
}
// static deal (k_tile): unit u goes to wave u mod (workgroups x 16)
template <typename T, int KQ>
__device__ __forceinline__ void leftover_units(const TileArgs<T>& a, const TileGeom& g, int lane, int wave) {
  const int n_units = a.P * g.n_left, n_waves = (int)gridDim.x * BSR_TILE_WAVES;
  for (int tk = wave * (int)gridDim.x + (int)blockIdx.x; tk < n_units; tk += n_waves) leftover_unit<T, KQ>(a, g, lane, tk);
}


}  // namespace
