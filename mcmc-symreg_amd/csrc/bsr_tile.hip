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
#include "bsr_tile_common.h"

namespace {


// The row pass.  Workgroup = (row slice, tape group); wave w of group g runs the up to QMAX tapes the host's schedule
// gives it (cost-balanced, sched[g][pass][w][q]) over the slice, with one set of per-lane sums per tape, and reduces
// them over the lanes together at the end (reduce_store).  The slice is staged in LDS whole where it fits (C2: 18
// columns x 4 blocks), else chunk by chunk through two buffers filled by LDS-DMA, the next chunk travelling while the
// waves run their tapes on this one (C5).
//
// A chain tape (bsr_device.h: chain_eval) runs a pass of NB blocks at a time, the whole pass in registers: one decode
// of the tape per pass, operators in place.  Any other tape goes through the stack machine two blocks at a time.
// Both produce the same values per row and add them up in the same order (per lane: blocks in order, the lane's two
// rows of a block in order), and what is summed in which order depends only on the context (slice boundaries), never
// on the batch, the schedule or the chunking: a proposal's partial sums -- hence its score -- are bit-identical
// whatever else shares the launch.
template <typename T, int KQ>
struct TileShape {
  // blocks per pass of a chain tape: four (16 VGPRs of values, as many of operand columns) next to the sums of K <= 4
  // basis columns, two beyond (K = 8: 22 VGPRs of sums per tape)
  static constexpr int NB = (sizeof(T) == 4 || KQ <= 4) ? BSR_TILE_NB : BSR_TILE_NB / 2;
  static constexpr int QMAX = BSR_TILE_QMAX;                   // tapes (sets of sums) per wave and pass
};

template <typename T, int KQ>
__global__ __launch_bounds__(BSR_TILE_WAVES* BSR_WAVE) void k_tile(TileArgs<T> a) {
  constexpr int U = BSR_TILE_U;
  constexpr int S = BSR_REG_STACK;
  constexpr int NB = TileShape<T, KQ>::NB;
  constexpr int QMAX = TileShape<T, KQ>::QMAX;
  using V2 = typename VecOf<T, 2>::type;
  extern __shared__ __align__(16) unsigned char smem[];
  T* sx = reinterpret_cast<T*>(smem);  // [buffers][ncols][chunk_rows]
  __shared__ double s_red[BSR_TILE_WAVES][QMAX * (KQ + 2) + 8];
  const TileGeom g = a.g;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tg = blockIdx.x / g.n_slices, slice = blockIdx.x - tg * g.n_slices;
  // slices of bps blocks (the first n_long of them one more); the blocks behind the last one: leftover_units
  const int b0 = slice * g.bps + min(slice, g.n_long), b1 = b0 + g.bps + (slice < g.n_long ? 1 : 0);
  const int sl_blocks = b1 - b0;
  const int chunk_rows = g.chunk_blocks * BSR_TILE_BLOCK;
  const bool multi = g.chunk_blocks < g.bps + (g.n_long > 0 ? 1 : 0);   // the slice does not fit LDS whole: two buffers, LDS-DMA
  const int buf_elems = (multi && sizeof(T) == 8) ? g.ncols * chunk_rows : 0;   // one buffer of the ring (f32: one buffer at all)
  const T* const CONSTANT_AS* colsrc = group_cols<T>(a, tg);
  const int y_slot = a.grp_nF[tg & 7];
  const int ncols = y_slot + 1 + (g.ncols_fixed);   // the group's LDS columns
  // diagnostics: shader-clock samples per wave (0 start, 1 first chunk staged, 2 first chunk computed, 3 all chunks
  // computed, 4 reductions stored; 7 and 6: the constant-rate 100 MHz clock at the start and at the end -- the shader
  // clock counters of different XCDs are not aligned, only differences inside one wave mean anything)
  unsigned long long* stamp = a.stamps ? a.stamps + ((size_t)blockIdx.x * BSR_TILE_WAVES + wave) * BSR_TILE_STAMP_WORDS : nullptr;
#define TSTAMP(i) do { if (stamp && lane == 0) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
  TSTAMP(0);
  if (stamp && lane == 0) stamp[7] = __builtin_amdgcn_s_memrealtime();
  tables_to_lds();  // visible after the first barrier below
  unsigned long long busy = 0, t_busy = 0;   // diagnostics: cycles the wave spent running tapes (not waiting for copies or barriers)
  for (int pass = 0; pass < g.n_pass; ++pass) {
    const TapeRec* my = a.sched + (((size_t)tg * g.n_pass + pass) * BSR_TILE_WAVES + wave) * QMAX;
    TapeAcc<KQ> A[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; ++q) A[q].clear();
    // f64 chunks travel through a ring of g.ring LDS buffers by LDS-DMA, ring - 1 chunks ahead of the one the waves
    // run their tapes on: what is in flight per CU (ring - 1 chunks, ~100 KB at C5) is what keeps HBM busy -- with one
    // chunk ahead the pass waited for memory three quarters of its time.  Copies complete in issue order, so "chunk ci
    // has landed" is a count of the wave's later copies (vmcnt); the barrier behind it also says everyone is done
    // with chunk ci - 1, whose buffer the next copies overwrite.
    constexpr bool DMA = sizeof(T) == 8;
    const int ring = (multi && DMA) ? g.ring : 1;
    const int n_chunks = (sl_blocks + g.chunk_blocks - 1) / g.chunk_blocks;
    int cnt[4] = {0, 0, 0, 0};   // copies this wave issued for chunk j, at j & 3
    // A wave copies the same (column, block of the chunk) units of every chunk: unit u = wave + 16 k.  Lane k keeps
    // unit k's source address (rows of chunk 0) and LDS offset, fetched ONCE from the column-pointer table -- a scalar
    // load of the pointer in front of every copy put a memory round trip between the barrier and the copies of
    // every chunk.
    const int n_units = ncols * g.chunk_blocks;
    const int n_mine = (multi && DMA && n_units > wave) ? min(BSR_WAVE, (n_units - wave + BSR_TILE_WAVES - 1) / BSR_TILE_WAVES) : 0;
    unsigned long long my_src = 0;
    int my_lds = 0, my_blk = 0;
    if (multi && DMA) {
      const int u = wave + BSR_TILE_WAVES * lane;
      if (u < n_units) {
        const int col = u / g.chunk_blocks;
        my_blk = u - col * g.chunk_blocks;
        my_src = (unsigned long long)(size_t)(colsrc[col] + (int64_t)(b0 + my_blk) * BSR_TILE_BLOCK);
        my_lds = (int)(((size_t)col * chunk_rows + (size_t)my_blk * BSR_TILE_BLOCK) * sizeof(T));
      }
    }
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)sx;
    auto issue = [&](int j) {
      if constexpr (DMA) {
        const int nbj = min(g.chunk_blocks, sl_blocks - j * g.chunk_blocks);   // blocks of chunk j
        const uint32_t buf = lds0 + (uint32_t)((size_t)(j % ring) * buf_elems * sizeof(T));
        const unsigned long long adv = (unsigned long long)j * g.chunk_blocks * BSR_TILE_BLOCK * sizeof(T);
        const uint32_t voff = (uint32_t)lane * 16u;
        int n = 0;
        for (int k = 0; k < n_mine; ++k) {
          if (__builtin_amdgcn_readlane(my_blk, k) >= nbj) continue;
          const unsigned long long src = (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(my_src >> 32), k) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)my_src, k)) + adv;
          const uint32_t la = buf + (uint32_t)__builtin_amdgcn_readlane(my_lds, k);
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(src), "s"(la) : "memory", "m0");
          ++n;
        }
        cnt[j & 3] = n;
      }
    };
    if (!multi) {
      if (pass == 0) {
        stage_rows<T, 8>(sx, colsrc, ncols, chunk_rows, b0, sl_blocks, wave, lane);
        __syncthreads();
        TSTAMP(1);
      }
    } else if (DMA) {
      if (pass != 0) __syncthreads();  // everyone is done with the last chunks of the pass before
      for (int j = 0; j < min(ring - 1, n_chunks); ++j) issue(j);
    }
    int ci = 0;
    for (int c0 = b0; c0 < b1; c0 += g.chunk_blocks, ++ci) {
      const int nbc = min(g.chunk_blocks, b1 - c0);
      const T* cur = sx + (size_t)(ci % ring) * buf_elems;
      if (multi) {
        if constexpr (DMA) {
          int later = 0;
          for (int j = ci + 1; j < min(ci + ring - 1, n_chunks); ++j) later += cnt[j & 3];
          dma_wait_left(later);
          __syncthreads();
          if (ci + ring - 1 < n_chunks) issue(ci + ring - 1);
          if (ci == 0 && pass == 0) TSTAMP(1);
        } else {
          // f32 columns keep one register-staged buffer (LDS-DMA moves 16 bytes per lane: half a block of floats)
          if (c0 != b0 || pass != 0) __syncthreads();  // everyone is done with the rows staged before
          stage_rows<T, 4>(sx, colsrc, ncols, chunk_rows, c0, nbc, wave, lane);
          __syncthreads();
          if (c0 == b0 && pass == 0) TSTAMP(1);
        }
      }
      const T* sy = cur + (size_t)y_slot * chunk_rows;
      if (stamp) t_busy = __builtin_amdgcn_s_memtime();
#pragma unroll 1
      for (int q = 0; q < QMAX; ++q) {
        const TapeRec CONSTANT_AS* rec = as_const(my + q);
        if (rec->p < 0) continue;
        const uint64_t* pc = a.codes + rec->code_off;
        const uint64_t* pf = a.feats + rec->feat_off;
        const double* pl = a.lnp + 2 * (size_t)rec->ln_off;
        const int n_nodes = rec->n_nodes;
        const double s = rec->s;
        const bool chain = (rec->chain & 1) != 0;   // (bit 1: the streaming kernel's own flag)
        const T* sq = cur + (size_t)rec->qslot * chunk_rows;
        TapeHead hd;
        hd.code0 = rec->code0; hd.code1 = rec->code1; hd.f0 = rec->f0; hd.f1 = rec->f1;
        hd.la = rec->ln[0]; hd.lb = rec->ln[1];
    hd.ln_near = (const double*)rec->ln;
        hd.n_ln = rec->n_ln;
        hd.n_term = rec->n_term;
        // the sums of block `blk` (absolute), whose lane pair sits at `off` of the staged rows, into set q
        auto add_block = [&](const T (&zz)[U], int off, int blk) {
          const int64_t row0 = (int64_t)blk * BSR_TILE_BLOCK + 2 * lane;
          const V2 yv = *reinterpret_cast<const V2*>(sy + off);
          V2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
          for (int i = 0; i < KQ; ++i) qv[i] = *reinterpret_cast<const V2*>(sq + (size_t)i * chunk_rows + off);
          // (every block of a slice lies below row N whole: the block that holds row N is a leftover unit -- the context's
          // geometry sees to it, bsr_api.hip -- so nothing in these loops masks rows: nine scalar instructions and a branch
          // per block at C2, a quarter of the kernel's scalar work)
#define BSR_ACC_CASE(qq)                                                               \
  case qq:                                                                             \
    if constexpr (qq < QMAX) {                                                         \
      accumulate_v<T, KQ, false>(A[qq], zz, yv, qv, s, row0, a.N);                     \
    }                                                                                  \
    break;
          switch (q) { BSR_ACC_CASE(0) BSR_ACC_CASE(1) BSR_ACC_CASE(2) BSR_ACC_CASE(3) }
#undef BSR_ACC_CASE
        };
        int b = 0;
        if (chain) {
          // a pass of PB blocks: the tape once over 2 PB values per lane, then the blocks' sums in order
          auto pass_nb = [&](auto nb_tag, auto full_tag, int pn) {
            constexpr int PB = decltype(nb_tag)::value;
            constexpr bool FULL = decltype(full_tag)::value;
            const int off = b * BSR_TILE_BLOCK + 2 * lane;  // the lane's pair inside the chunk (block j of the pass: 128 j rows on)
            T z[2 * PB];
            chain_eval<T, PB, FULL>(hd, pc, pf, pl, n_nodes, cur, chunk_rows, off, pn, z);
#pragma unroll
            for (int jb = 0; jb < PB; ++jb) {
              if (FULL || jb < pn) {
                const T zz[U] = {z[2 * jb], z[2 * jb + 1]};
                add_block(zz, off + jb * BSR_TILE_BLOCK, c0 + b + jb);
              }
            }
            b += pn;
          };
          using std::integral_constant;
          while (nbc - b >= NB) pass_nb(integral_constant<int, NB>{}, integral_constant<bool, true>{}, NB);
          // (chunks of two blocks -- wide groups at N = 1M -- and the tails of others: a pass of two, whole)
          if (NB > 2 && nbc - b >= 2) pass_nb(integral_constant<int, 2>{}, integral_constant<bool, true>{}, 2);
          if (b < nbc) pass_nb(integral_constant<int, 2>{}, integral_constant<bool, false>{}, nbc - b);
        } else {
          // Two blocks per interpreter pass (4 rows per lane): the scalar decode of a node is paid once per 256 rows.
#pragma unroll 1
          for (; b + 1 < nbc; b += 2) {
            const int off = b * BSR_TILE_BLOCK + 2 * lane;  // the lane's pair inside the chunk (second pair 128 rows on)
            T z4[2 * U];
            LdsCols<T, 2 * U> ldr{cur, chunk_rows, off};
            run_tape_head<T, 2 * U, S, LdsCols<T, 2 * U>, false>(hd, pc, pf, pl, n_nodes, ldr, z4, (T*)nullptr, lane);
            const T za[U] = {z4[0], z4[1]}, zb[U] = {z4[2], z4[3]};
            add_block(za, off, c0 + b);
            add_block(zb, off + BSR_TILE_BLOCK, c0 + b + 1);
          }
          if (b < nbc) {
            const int off = b * BSR_TILE_BLOCK + 2 * lane;
            T z[U];
            LdsCols<T, U> ldr{cur, chunk_rows, off};
            run_tape_head<T, U, S, LdsCols<T, U>, false>(hd, pc, pf, pl, n_nodes, ldr, z, (T*)nullptr, lane);
            add_block(z, off, c0 + b);
          }
        }
      }
      if (stamp) busy += __builtin_amdgcn_s_memtime() - t_busy;
      if (c0 == b0 && pass == 0) TSTAMP(2);
    }
    if (pass == g.n_pass - 1) TSTAMP(3);
    if (stamp && lane == 0) stamp[5] = busy;
    reduce_store<KQ, QMAX>(A, my, a.part, g.n_part, slice, s_red[wave], lane);
  }
  if (g.n_left > 0) leftover_units<T, KQ>(a, g, lane, wave);
  TSTAMP(4);
  if (stamp && lane == 0) stamp[6] = __builtin_amdgcn_s_memrealtime();
#undef TSTAMP
}

// Whole-slice variant with tape pulling (contexts whose slices fit LDS whole: C2, C3, C4).  The slice is staged once;
// then the waves PULL tapes from the group's list (heaviest first, a.glist) through an LDS counter, and a wave runs
// its tape over all blocks of the slice with one set of sums: with two to four tapes per wave of very different cost
// (a leaf: ~230 vector instructions per slice, a nest of three transcendentals: ~1 900) pulling balances what a
// static schedule cannot, and a slice twice as long (T = 2 groups) halves the per-tape costs: decode, record fetch,
// lane reduction.  Which wave runs a tape does not matter to the sums (one wave, blocks in order).  Chain tapes run a
// pass of up to eight blocks at a time (K <= 4), the whole pass in registers.
template <typename T, int KQ>
__global__ __launch_bounds__(BSR_TILE_WAVES* BSR_WAVE) void k_tile1(TileArgs<T> a) {
  constexpr int U = BSR_TILE_U;
  constexpr int S = BSR_REG_STACK;
  constexpr int NBIG = (sizeof(T) == 4 || KQ <= 4) ? 8 : 4;   // blocks per pass of a chain tape: what the registers hold
  using V2 = typename VecOf<T, 2>::type;
  extern __shared__ __align__(16) unsigned char smem[];
  T* sx = reinterpret_cast<T*>(smem);  // [ncols][chunk_rows]
  __shared__ int s_next;
  const TileGeom g = a.g;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tg = blockIdx.x / g.n_slices, slice = blockIdx.x - tg * g.n_slices;
  // slices of bps whole blocks, the first n_long of them one more (LDS holds the longest: chunk_blocks per column)
  const int b0 = slice * g.bps + min(slice, g.n_long);
  const int nb = g.bps + (slice < g.n_long ? 1 : 0);
  const int chunk_rows = g.chunk_blocks * BSR_TILE_BLOCK;
  const T* const CONSTANT_AS* colsrc = group_cols<T>(a, tg);
  const int y_slot = a.grp_nF[tg & 7];
  const int ncols = y_slot + 1 + g.ncols_fixed;
  const T* sy = sx + (size_t)y_slot * chunk_rows;
  // the group's tapes in cost order: indices of their records
  const int32_t CONSTANT_AS* list = as_const(reinterpret_cast<const int32_t*>(a.sched + (size_t)g.T * g.n_pass * BSR_TILE_WAVES * g.qmax) +
                                             a.P + (size_t)tg * g.per_group);
  unsigned long long* stamp = a.stamps ? a.stamps + ((size_t)blockIdx.x * BSR_TILE_WAVES + wave) * BSR_TILE_STAMP_WORDS : nullptr;
#define TSTAMP(i) do { if (stamp && lane == 0) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
  TSTAMP(0);
  if (stamp && lane == 0) stamp[7] = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) s_next = BSR_TILE_WAVES;
  // the slice's columns first, the math tables behind them: one round trip to memory for both (the tables in front,
  // with their own wait before the LDS write, cost the staging a second one)
  if constexpr (sizeof(T) == 8) {
    dma_rows<T>(sx, colsrc, ncols, chunk_rows, b0, nb, wave, lane);
    for (int t = wave; t < (int)(BSR_TAB_DOUBLES * sizeof(double) / 1024); t += BSR_TILE_WAVES) {
      const char* src = (const char*)bsr_tables_src + t * 1024 + lane * 16;
      const uint32_t la = __builtin_amdgcn_readfirstlane(
          (uint32_t)(size_t)(__attribute__((address_space(3))) void*)((char*)bsr_lds_tab + t * 1024));
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(la) : "memory", "m0");
    }
    dma_wait();
  } else {
    stage_rows<T, 8>(sx, colsrc, ncols, chunk_rows, b0, nb, wave, lane);
    tables_to_lds();
  }
  __syncthreads();
  TSTAMP(1);
  TSTAMP(2);
  const int n_items = g.per_group;
  int idx = wave;   // the first list entries go to the waves in order
  while (idx < n_items) {
    const int ri = list[idx];
    if (ri < 0) {   // padding behind the group's last tape
      int nx = 0;
      if (lane == 0) nx = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      idx = __builtin_amdgcn_readfirstlane(nx);
      continue;
    }
    // the next unit is requested now; its round trip hides under this one
    int nxt = 0;
    if (lane == 0) nxt = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // The list is in cost order: a heavy tape on an even share of its SIMD's issue slots (four waves) would end long
    // after the list has drained; the first tapes of the list run at raised priority instead.
    if (idx < 4) __builtin_amdgcn_s_setprio(3);
    else if (idx < 8) __builtin_amdgcn_s_setprio(2);
    else if (idx < BSR_TILE_WAVES) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    const TapeRec CONSTANT_AS* rec = as_const(a.sched + ri);
    const int p = rec->p;
    const uint64_t* pc = a.codes + rec->code_off;
    const uint64_t* pf = a.feats + rec->feat_off;
    const double* pl = a.lnp + 2 * (size_t)rec->ln_off;
    const int n_nodes = rec->n_nodes;
    const double s = rec->s;
    const bool chain = (rec->chain & 1) != 0;   // (bit 1: the streaming kernel's own flag)
    const T* sq = sx + (size_t)rec->qslot * chunk_rows;
    TapeHead hd;
    hd.code0 = rec->code0; hd.code1 = rec->code1; hd.f0 = rec->f0; hd.f1 = rec->f1;
    hd.la = rec->ln[0]; hd.lb = rec->ln[1];
    hd.ln_near = (const double*)rec->ln;
    hd.n_ln = rec->n_ln;
    hd.n_term = rec->n_term;
    TapeAcc<KQ> A;
    A.clear();
    // one 128-row block: the lane's pair of rows, sums in row order
    auto add_block = [&](const T (&zz)[U], int off, int b) {
      const int64_t row0 = (int64_t)(b0 + b) * BSR_TILE_BLOCK + 2 * lane;
      const V2 yv = *reinterpret_cast<const V2*>(sy + off);
      V2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
      for (int i = 0; i < KQ; ++i) qv[i] = *reinterpret_cast<const V2*>(sq + (size_t)i * chunk_rows + off);
      accumulate_v<T, KQ, false>(A, zz, yv, qv, s, row0, a.N);   // (slices hold whole blocks only: see k_tile)
    };
    int b = 0;
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
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          if (FULL || jb < pn) {
            const T zz[U] = {z[2 * jb], z[2 * jb + 1]};
            const V2 yv = *reinterpret_cast<const V2*>(yp + jb * BSR_TILE_BLOCK);
            V2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
            for (int i = 0; i < KQ; ++i) qv[i] = *reinterpret_cast<const V2*>(qp + (size_t)i * chunk_rows + jb * BSR_TILE_BLOCK);
            const int64_t row0 = (int64_t)(b0 + b + jb) * BSR_TILE_BLOCK + 2 * lane;
            accumulate_v<T, KQ, false>(A, zz, yv, qv, s, row0, a.N);
          }
        }
        b += pn;
      };
      using std::integral_constant;
      while (NBIG > 4 && nb - b >= NBIG) pass(integral_constant<int, NBIG>{}, integral_constant<bool, true>{}, NBIG);
      while (nb - b >= 4) pass(integral_constant<int, 4>{}, integral_constant<bool, true>{}, 4);
      // (what is left: a full pass of two blocks costs two blocks; a partial pass of four would cost four)
      if (nb - b >= 2) pass(integral_constant<int, 2>{}, integral_constant<bool, true>{}, 2);
      if (b < nb) pass(integral_constant<int, 2>{}, integral_constant<bool, false>{}, nb - b);
    } else {
      // Two blocks per interpreter pass (4 rows per lane): the scalar decode of a node is paid once per 256 rows.  The
      // per-lane sums still grow block by block in row order, so the result does not depend on the pairing.
#pragma unroll 1
      for (; b + 1 < nb; b += 2) {
        const int off = b * BSR_TILE_BLOCK + 2 * lane;  // the lane's pair inside the slice (second pair 128 rows on)
        T z4[2 * U];
        LdsCols<T, 2 * U> ldr{sx, chunk_rows, off};
        // four values per out-of-line call, except where that costs the last registers (K = 6, 8 would spill two)
        run_tape_head<T, 2 * U, S, LdsCols<T, 2 * U>, (sizeof(T) == 4 || (KQ != 6 && KQ != 8))>(hd, pc, pf, pl, n_nodes, ldr, z4, (T*)nullptr, lane);
        const T za[U] = {z4[0], z4[1]}, zb[U] = {z4[2], z4[3]};
        add_block(za, off, b);
        add_block(zb, off + BSR_TILE_BLOCK, b + 1);
      }
      if (b < nb) {
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
  // the workgroup's share of the leftover units (unit u of the launch: workgroup u mod gridDim) through the same
  // counter: the waves that run out of tapes first take them, heaviest tape first
  if (g.n_left > 0) {
    __builtin_amdgcn_s_setprio(0);
    const int n_units = a.P * g.n_left;
    for (;;) {
      const int tk = (idx - n_items) * (int)gridDim.x + (int)blockIdx.x;
      if (tk >= n_units) break;
      int nx = 0;
      if (lane == 0) nx = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      leftover_unit<T, KQ>(a, g, lane, tk);
      idx = __builtin_amdgcn_readfirstlane(nx);
    }
  }
  TSTAMP(4);
  if (stamp && lane == 0) stamp[6] = __builtin_amdgcn_s_memrealtime();
#undef TSTAMP
}

template <typename T, int KQ>
void launch_kq(hipStream_t st, const TileArgs<T>& a) {
  const TileGeom& g = a.g;
  const dim3 grid((unsigned)(g.T * g.n_slices)), block(BSR_TILE_WAVES * BSR_WAVE);
  // a slice that does not fit whole travels through two buffers (fp64: LDS-DMA double buffering)
  const bool multi = g.chunk_blocks < g.bps + (g.n_long > 0 ? 1 : 0);
  const size_t lds = (size_t)g.ncols * g.chunk_blocks * BSR_TILE_BLOCK * sizeof(T) * ((multi && sizeof(T) == 8) ? g.ring : 1);
  if constexpr (sizeof(T) == 8) if (g.per_group > 0) {
    static bool attr1 = false;
    if (!attr1) {
      (void)hipFuncSetAttribute((const void*)k_tile1<T, KQ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(tile_lds_bytes_max() - 1024));
      attr1 = true;
    }
    bsr_launch((k_tile1<T, KQ>), grid, block, lds, st, a);
    return;
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_tile<T, KQ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(tile_lds_bytes_max() - 1024));
    attr = true;
  }
  bsr_launch((k_tile<T, KQ>), grid, block, lds, st, a);
}

}  // namespace

int tile_qmax(int K) { (void)K; return BSR_TILE_QMAX; }

// dynamic LDS a launch may ask for: the math tables and the reduction scratch are static LDS
size_t tile_lds_bytes_max() { return 160 * 1024 - BSR_TAB_DOUBLES * sizeof(double) - 4096; }

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
