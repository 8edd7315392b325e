// The whole-slice row pass with its tape loop in gfx950 assembly (fp64, K <= 4): the scoring pass of the headline
// configuration (allcal + the O(N) part of ylogLike / the rank gate, codes/funcs.py:175-220, 1147-1174, 1212-1226).
//
// Same geometry, same staging, same sums and the same partial records -- byte for byte -- as bsr_tile.hip: k_tile1; what
// differs is who runs the tapes.  Each wave pulls tapes from the group's list; the ones the host packed as 64-byte programs
// (TileProg: chain tapes of at most 17 entries) run inside ONE block of assembly (bsr_tile_asm.h): program by a single
// scalar load, threaded-code dispatch (5 scalar instructions per entry and pass of four blocks), the pass's rows into the
// sums, lane reduction, record store, next tape.  The block hands back what it does not take -- tapes for the stack
// machine, a sin / cos with a huge or non-finite argument -- and the C++ interpreter of bsr_device.h runs those exactly
// as k_tile1 would.
//
// Split staging: the slice's first four blocks of every column are requested first, the rest behind them; the waves
// start on the first half when IT has landed and meet at a second barrier before anyone touches the second half
// (copies complete in issue order: s_waitcnt vmcnt(n) with n = the wave's copies of the second half).
#include "bsr_tile_common.h"
#include "bsr_tile_asm.h"

namespace {

// One tape through the C++ interpreter over the whole slice (what k_tile1 does for every tape), out of line: its
// registers are not the tape loop's.
// (a pointer argument of a call arrives in vector registers: made scalar by hand, so that what hangs off it is read by
// scalar loads)
template <typename P>
__device__ __forceinline__ const P CONSTANT_AS* uniform_const(const void* q) {
  const uint64_t v = (uint64_t)(size_t)q;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const P CONSTANT_AS*)(size_t)(((uint64_t)hi << 32) | lo);
}

template <int KQ>
__device__ __attribute__((noinline)) void tile_tape_cpp(const TileArgs<double>* ka, const TapeRec* recp,
                                                        const double* sx, int chunk_rows, int y_slot, int b0, int nb,
                                                        int slice, int lane) {
  using T = double;
  constexpr int U = BSR_TILE_U;
  constexpr int S = BSR_REG_STACK;
  using V2 = typename VecOf<T, 2>::type;
  const TileArgs<double> CONSTANT_AS* ap = uniform_const<TileArgs<double>>(ka);
  const TapeRec CONSTANT_AS* rec = uniform_const<TapeRec>(recp);
  chunk_rows = __builtin_amdgcn_readfirstlane(chunk_rows);
  y_slot = __builtin_amdgcn_readfirstlane(y_slot);
  b0 = __builtin_amdgcn_readfirstlane(b0);
  nb = __builtin_amdgcn_readfirstlane(nb);
  slice = __builtin_amdgcn_readfirstlane(slice);
  const double* sy = sx + (size_t)y_slot * chunk_rows;
  const int p = rec->p;
  const uint64_t* pc = ap->codes + rec->code_off;
  const uint64_t* pf = ap->feats + rec->feat_off;
  const double* pl = ap->lnp + 2 * (size_t)rec->ln_off;
  const int n_nodes = rec->n_nodes;
  const double s = rec->s;
  const bool chain = (rec->chain & 1) != 0;
  const T* sq = sx + (size_t)rec->qslot * chunk_rows;
  TapeHead hd;
  hd.code0 = rec->code0; hd.code1 = rec->code1; hd.f0 = rec->f0; hd.f1 = rec->f1;
  hd.la = rec->ln[0]; hd.lb = rec->ln[1];
  hd.ln_near = (const double*)rec->ln;
  hd.n_ln = rec->n_ln;
  hd.n_term = rec->n_term;
  TapeAcc<KQ> A;
  A.clear();
  const int64_t N = ap->N;
  auto add_block = [&](const T (&zz)[U], int off, int b) {
    const int64_t row0 = (int64_t)(b0 + b) * BSR_TILE_BLOCK + 2 * lane;
    const V2 yv = *reinterpret_cast<const V2*>(sy + off);
    V2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
    for (int i = 0; i < KQ; ++i) qv[i] = *reinterpret_cast<const V2*>(sq + (size_t)i * chunk_rows + off);
    accumulate_v<T, KQ, false>(A, zz, yv, qv, s, row0, N);
  };
  int b = 0;
  if (chain) {
    // passes of two blocks: the values of a row and the order of every sum do not depend on the pass length
#pragma unroll 1
    while (b < nb) {
      const int pn = min(2, nb - b);
      const int off = b * BSR_TILE_BLOCK + 2 * lane;
      T z[4];
      if (pn == 2) chain_eval<T, 2, true>(hd, pc, pf, pl, n_nodes, sx, chunk_rows, off, 2, z);
      else chain_eval<T, 2, false>(hd, pc, pf, pl, n_nodes, sx, chunk_rows, off, 1, z);
      const T za[U] = {z[0], z[1]}, zb[U] = {z[2], z[3]};
      add_block(za, off, b);
      if (pn == 2) add_block(zb, off + BSR_TILE_BLOCK, b + 1);
      b += pn;
    }
  } else {
#pragma unroll 1
    for (; b < nb; ++b) {
      const int off = b * BSR_TILE_BLOCK + 2 * lane;
      T z[U];
      LdsCols<T, U> ldr{sx, chunk_rows, off};
      run_tape_head<T, U, S>(hd, pc, pf, pl, n_nodes, ldr, z, (T*)nullptr, lane);
      add_block(z, off, b);
    }
  }
  store_partial<KQ>(A, ap->part + ((size_t)p * ap->g.n_part + slice) * BSR_P1_WORDS, lane);
}

template <int KQ>
__device__ __attribute__((noinline)) void tile_leftover_cpp(const TileArgs<double>* ka, int lane, int tk) {
  const TileArgs<double> CONSTANT_AS* ap = uniform_const<TileArgs<double>>(ka);
  tk = __builtin_amdgcn_readfirstlane(tk);
  leftover_unit<double, KQ, TileArgs<double> CONSTANT_AS, TileGeom CONSTANT_AS>(*ap, ap->g, lane, tk);
}

template <int KQ>
__global__ __launch_bounds__(BSR_TILE_WAVES* BSR_WAVE) void k_tile1a(TileArgs<double> a) {
  static_assert(KQ >= 1 && KQ <= 4, "the tape loop's block is written for one to four basis columns");
  using T = double;
  extern __shared__ __align__(16) unsigned char smem[];
  T* sx = reinterpret_cast<T*>(smem);  // [ncols][chunk_rows]
  __shared__ int s_next;
  const TileArgs<double>* ap = (const TileArgs<double>*)__builtin_amdgcn_kernarg_segment_ptr();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n_slices = a.g.n_slices;
  const int tg = blockIdx.x / n_slices, slice = blockIdx.x - tg * n_slices;
  // slices of bps whole blocks, the first n_long of them one more (LDS holds the longest: chunk_blocks per column)
  const int nb = a.g.bps + (slice < a.g.n_long ? 1 : 0);
  const int b0 = slice * a.g.bps + min(slice, a.g.n_long);
  const int chunk_rows = a.g.chunk_blocks * BSR_TILE_BLOCK;
  const T* const CONSTANT_AS* colsrc = group_cols<T>(a, tg);
  const int y_slot = a.grp_nF[tg & 7];
  const int ncols = y_slot + 1 + a.g.ncols_fixed;
  unsigned long long* stamp = a.stamps ? a.stamps + ((size_t)blockIdx.x * BSR_TILE_WAVES + wave) * BSR_TILE_STAMP_WORDS : nullptr;
#define TSTAMP(i) do { if (stamp && lane == 0) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
  TSTAMP(0);
  if (stamp && lane == 0) {
    stamp[7] = __builtin_amdgcn_s_memrealtime();
    // where the wave runs: HW_ID (cu_id [11:8], sh_id [12], se_id [15:13], simd_id [5:4]) and XCC_ID [3:0] -- the busy
    // intervals of every CU from the stamps of many launches (tools/cu_occupancy.py)
    stamp[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
               ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  }
  if (threadIdx.x == 0) s_next = BSR_TILE_WAVES;
  // Staging by LDS-DMA in two halves: blocks [0, 4) of every column and the math tables; when those have landed for
  // everyone, blocks [4, nb) are requested and the waves start on the first half.  (Requested all at once, a wave's
  // first-half copies queue behind the other waves' second-half ones in the CU's memory pipeline: the first barrier
  // then waits for nearly everything.)  (column, block) units of a half are dealt to the waves round robin.
  const int nb_a = a.split_stage ? min(nb, 4) : nb, nb_b = nb - nb_a;
  auto stage_half = [&](int first, int count) {
    const int n_u = ncols * count;
    for (int u = wave; u < n_u; u += BSR_TILE_WAVES) {
      const int col = u / count, blk = first + (u - col * count);
      const T* src = colsrc[col] + (int64_t)(b0 + blk) * BSR_TILE_BLOCK + 2 * lane;
      T* dst = sx + (size_t)col * chunk_rows + (size_t)blk * BSR_TILE_BLOCK;
      const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)dst);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(la) : "memory", "m0");
    }
  };
  stage_half(0, nb_a);
  for (int t = wave; t < (int)(BSR_TAB_DOUBLES * sizeof(double) / 1024); t += BSR_TILE_WAVES) {
    const char* src = (const char*)bsr_tables_src + t * 1024 + lane * 16;
    const uint32_t la = __builtin_amdgcn_readfirstlane(
        (uint32_t)(size_t)(__attribute__((address_space(3))) void*)((char*)bsr_lds_tab + t * 1024));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(la) : "memory", "m0");
  }
  dma_wait();
  __syncthreads();
  uint32_t pend = nb_b > 0 ? 1u : 0u;
  if (nb_b > 0) stage_half(nb_a, nb_b);
  TSTAMP(1);
  TSTAMP(2);
  auto second_half = [&]() {   // the wave's copies of the second half, then everyone's
    if (pend) {
      dma_wait();
      __syncthreads();
      pend = 0;
    }
  };

  const int n_items = a.g.per_group;
  const int32_t CONSTANT_AS* list = as_const(reinterpret_cast<const int32_t*>(a.sched + (size_t)a.g.T * a.g.n_pass * BSR_TILE_WAVES * a.g.qmax) +
                                             a.P + (size_t)tg * n_items);
  const TileProg* progs = a.tprog + (size_t)tg * (n_items + 1);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)sx;
  const uint32_t lc = lds0 + (uint32_t)lane * 16u;
  const uint32_t stride = (uint32_t)chunk_rows * 8u;
  const uint32_t yoff = (uint32_t)y_slot * stride;
  const uint32_t tab_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)bsr_lds_tab;
  const uint32_t snext_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&s_next;
  const uint64_t part0 = (uint64_t)(size_t)(a.part + (size_t)slice * BSR_P1_WORDS);
  const uint32_t part_lo = (uint32_t)part0, part_hi = (uint32_t)(part0 >> 32);
  const uint32_t np96 = (uint32_t)a.g.n_part * (uint32_t)(BSR_P1_WORDS * sizeof(double));
  // which word of the record the lanes 0, 16, 32, 48 write (the rows of the reduced register), as byte offsets
  const int row = lane >> 4;
  uint32_t so, so2 = (lane >= 32) ? 72u : 64u;
  uint64_t m4 = 0x0001000100010001ull;
  const uint64_t m2 = 0x0000000100000001ull;
  if constexpr (KQ == 1) { so = row == 0 ? 0u : row == 1 ? 72u : 64u; m4 = 0x0000000100010001ull; }
  else if constexpr (KQ == 2) so = row == 0 ? 0u : row == 1 ? 64u : row == 2 ? 8u : 72u;
  else if constexpr (KQ == 3) so = row == 0 ? 0u : row == 1 ? 16u : row == 2 ? 8u : 64u;
  else so = row == 0 ? 0u : row == 1 ? 16u : row == 2 ? 8u : 24u;
  const uint32_t bps_u = (uint32_t)nb, nitems_u = (uint32_t)n_items;

  uint32_t idx = (uint32_t)wave, nxt = 0, st;
  uint32_t idx_v = idx, nxt_v = 0, pend_v = pend;
#pragma clang loop unroll(disable)
  for (;;) {
#define BSR_TA_OPERANDS                                                                                                \
    : [idxv] "+v"(idx_v), [nxtv] "=v"(nxt_v), [st] "=s"(st), [pendv] "+v"(pend_v)                                      \
    : [progs] "s"(progs), [nitems] "s"(nitems_u), [stride] "s"(stride), [bps] "s"(bps_u), [part] "s"(part_lo),          \
      [parth] "s"(part_hi), [np96] "s"(np96), [yoff] "s"(yoff), [tab] "s"(tab_lds), [snext] "s"(snext_lds),             \
      [lc] "v"(lc), [so] "v"(so), [so2] "v"(so2), [m4] "s"(m4), [m2] "s"(m2)                                            \
    : BSR_TILE_TAPES_CLOBBERS
    if constexpr (KQ == 1) asm volatile(BSR_TILE_TAPES_ASM_K1 BSR_TA_OPERANDS);
    else if constexpr (KQ == 2) asm volatile(BSR_TILE_TAPES_ASM_K2 BSR_TA_OPERANDS);
    else if constexpr (KQ == 3) asm volatile(BSR_TILE_TAPES_ASM_K3 BSR_TA_OPERANDS);
    else asm volatile(BSR_TILE_TAPES_ASM_K4 BSR_TA_OPERANDS);
#undef BSR_TA_OPERANDS
    idx = __builtin_amdgcn_readfirstlane(idx_v);
    nxt = __builtin_amdgcn_readfirstlane(nxt_v);
    pend = __builtin_amdgcn_readfirstlane(pend_v);
    if (st == 0) break;
    // a tape for the C++ interpreter: list entry idx; the block has already pulled the index behind it
    second_half();
    tile_tape_cpp<KQ>(ap, a.sched + list[idx], sx, chunk_rows, y_slot, b0, nb, slice, lane);
    idx_v = nxt;
    pend_v = pend;
  }
  second_half();
  TSTAMP(3);
  // the workgroup's share of the leftover units (unit u of the launch: workgroup u mod gridDim) through the same
  // counter: the waves that run out of tapes first take them, heaviest tape first
  if (a.g.n_left > 0) {
    const int n_units = a.P * a.g.n_left;
    int li = (int)idx;
    for (;;) {
      const int tk = (li - n_items) * (int)gridDim.x + (int)blockIdx.x;
      if (tk >= n_units) break;
      int nx = 0;
      if (lane == 0) nx = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      tile_leftover_cpp<KQ>(ap, lane, tk);
      li = __builtin_amdgcn_readfirstlane(nx);
    }
  }
  TSTAMP(4);
  if (stamp && lane == 0) stamp[6] = __builtin_amdgcn_s_memrealtime();
#undef TSTAMP
}

template <int KQ>
void launch_kq_asm(hipStream_t st, const TileArgs<double>& a) {
  const TileGeom& g = a.g;
  const dim3 grid((unsigned)(g.T * g.n_slices)), block(BSR_TILE_WAVES * BSR_WAVE);
  const size_t lds = (size_t)g.ncols * g.chunk_blocks * BSR_TILE_BLOCK * sizeof(double);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_tile1a<KQ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(tile_lds_bytes_max() - 1024));
    attr = true;
  }
  bsr_launch((k_tile1a<KQ>), grid, block, lds, st, a);
}

}  // namespace

bool tile_asm_takes(int K) { return K >= 1 && K <= 4; }

void launch_tile_asm(hipStream_t st, const TileArgs<double>& a) {
  switch (a.K) {
    case 1: launch_kq_asm<1>(st, a); break;
    case 2: launch_kq_asm<2>(st, a); break;
    case 3: launch_kq_asm<3>(st, a); break;
    default: launch_kq_asm<4>(st, a); break;
  }
}
