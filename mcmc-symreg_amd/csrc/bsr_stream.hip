// Streaming row pass for data sets far larger than LDS (BASELINE configs[4]: N = 1M, d = 50): the projection pass of the
// scoring path (allcal + the O(N) part of ylogLike / the rank gate, codes/funcs.py:175-220, 1147-1174, 1212-1226) for
// fp64 contexts whose row slices do not fit LDS whole.
//
// One workgroup of 16 waves per CU owns a row slice and one tape group; a wave keeps the per-lane sums of its QT tapes
// (a static, cost-balanced schedule written by the host) in registers over the whole slice, and the slice travels
// HBM -> LDS through a ring of R buffers filled by LDS-DMA, R - 1 chunks ahead of the one the waves compute on.  What
// k_tile (bsr_tile.hip) paid per (tape, chunk) and this kernel does not:
//   * a tape's program -- 4-bit opcodes, 8-bit LDS slots of its terminals, prescale, basis slot -- sits in scalar
//     registers for the life of the wave (k_tile re-read a 128-byte record per tape and chunk, and its compiler-made
//     scalar spills were vector instructions: 34 per (tape, 128 rows) for one fused operand; tools/probes/op_costs);
//   * y and the basis columns are read from LDS once per chunk and wave, not once per tape, and before the tapes run;
//   * the operand of the next `acc op= column` entry is requested one entry ahead;
//   * a wave's DMA pieces take their source from scalar registers (a column base per piece, loaded once) and ONE vector
//     offset that advances by a chunk: no per-piece lane reads.
// Chunks are one or two 128-row blocks.  With all 64 tapes in ONE group (four per wave at K <= 4) every column is
// streamed once per launch (k_tile's two groups both streamed the columns they shared: 1.38x the algorithmic bytes).
//
// The fast interpreter exists three times over (template parameter MODE; BSR_STREAM_ASM): in C++ (tape_fast below), in
// gfx950 assembly a tape at a time (bsr_stream_asm.h), and as one block of assembly that runs a wave's tapes of a chunk --
// and, in mode 3, the loop over the slice's chunks -- (bsr_stream_chunk_asm.h: the default; a CU has one scalar unit for
// its sixteen waves, and the compiler's interpreter is a scalar program).  All three give a row's value the same bits.
//
// Tapes the fast interpreter does not take (more than 16 entries, 8 terminals or 3 ln nodes, more than one value below
// the accumulator, a `log`) go to the stack machine of bsr_device.h on the same staged rows, out of line (generic_block).
// Nothing else in the loop is a call (sin, cos, exp inline; only sin / cos of huge arguments go out of line): every device
// function starts with s_waitcnt vmcnt(0), which waits for the wave's copies in flight.  Either way the values of a row
// and the order of every sum are those of the other row passes: per lane the blocks of the slice in order, the lane's two
// rows of a block in order; one lane reduction per (tape, slice) in the fixed network of reduce_store.  What is summed in
// which order depends on the context's slices only, never on the batch.
#include "bsr_tile_common.h"
#include "bsr_stream_asm.h"
#include "bsr_stream_chunk_asm.h"

int env_int(const char* name, int dflt);   // bsr_api.hip
int stream_qmax(int K);

namespace {

// the scalar state of one tape of a wave: a StreamRec (bsr_internal.h), 32 bytes the host packs per (wave, set of sums),
// fetched by ONE scalar load
struct TapeS {
  uint32_t meta;        // bits 0..4: stream entries - 1 (a fast tape holds at most 16), bit 5 and bit 31: the fast
                        // interpreter takes it, bit 6: there is a tape, bits 8..15: LDS slot of the chain's first basis column
  uint32_t first;       // byte offset of the leading terminal's column in a chunk buffer (LDS slot x 1024 x blocks per chunk)
  double s;             // prescale
  uint64_t code;        // the entries behind the leading terminal, 4 bits each: operator + 1, 0 = end (bsr_stream_asm.h)
  uint64_t slots;       // LDS slots of the terminals behind the first, in stream order, 8 bits each
  __device__ __forceinline__ bool fast() const { return (meta & 32) != 0; }
  __device__ __forceinline__ bool any() const { return (meta & 64) != 0; }
  __device__ __forceinline__ int n() const { return (int)(meta & 31) + 1; }
  __device__ __forceinline__ int qslot() const { return (int)((meta >> 8) & 0xFF); }
};
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ TapeS load_tape(const StreamRec CONSTANT_AS* r) {
  const u32x8 w = *reinterpret_cast<const u32x8 CONSTANT_AS*>(r);
  TapeS t;
  t.meta = w[0];
  t.first = w[1];
  t.s = __hiloint2double((int)w[3], (int)w[2]);
  t.code = ((uint64_t)w[5] << 32) | w[4];
  t.slots = ((uint64_t)w[7] << 32) | w[6];
  return t;
}

template <int CB>
__device__ __forceinline__ void lds_pairs(const double* col, double (&v)[2 * CB]) {
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const double2 p = *reinterpret_cast<const double2*>(col + j * BSR_TILE_BLOCK);
    v[2 * j] = p.x;
    v[2 * j + 1] = p.y;
  }
}

// acc <- log|acc| through the out-of-line routine of bsr_device.h (an extension operator: the one call left in the loop)
#define BSR_STREAM_CALL(f)                                                              \
  if constexpr (CB == 2) {                                                              \
    double4 v = make_double4(acc[0], acc[1], acc[2], acc[3]);                           \
    v = f(v);                                                                           \
    acc[0] = v.x; acc[1] = v.y; acc[2] = v.z; acc[3] = v.w;                             \
  } else {                                                                              \
    double2 v = make_double2(acc[0], acc[1]);                                           \
    v = f(v);                                                                           \
    acc[0] = v.x; acc[1] = v.y;                                                         \
  }

// A tape on the lane's rows of one chunk, from scalar registers: entry 0 loads the leading terminal; `acc op= column`
// entries and unary operators map the accumulator to the accumulator (a chain tape holds nothing else); a terminal that
// is not fused pushes the accumulator into ONE saved register set, a binary operator pops it -- expressions like
// (x0 + x1) * (x2 + x3) cost two moves more than a chain, not a trip through the general stack machine (measured at
// ten times a chain's time per tape: tools/probes/op_costs).  Same operators, same operand order as run_tape_head.
// `lane_col`: the lane's pair in block 0 of column 0 of the chunk's buffer; `ln_tab`: the tape's (a, b) pairs in LDS.
template <int CB>
__device__ __forceinline__ void tape_fast(const TapeS& t, const double* lane_col, int chunk_rows, const double2* ln_tab,
                                          double (&acc)[2 * CB]) {
  constexpr int U = 2 * CB;
  uint64_t code = t.code, sl = t.slots;   // (the leading terminal's values arrive in acc)
  // (no operand requested ahead of its entry: carried around the loop, the compiler copies such registers on every
  // entry and waits for the read it was meant to hide -- 450 cycles per entry measured; at its use the read costs one LDS
  // round trip, which the SIMD's other waves fill)
  double s0[U];
#pragma unroll
  for (int u = 0; u < U; ++u) s0[u] = 0.0;
  int lk = 0;
  const int n = t.n();
  for (int i = 1; i < n; ++i) {
    const int op = (int)(code & 15u) - 1;
    code >>= 4;
    switch (op) {
      case BSR_SOP_ADD_T: {
        double pre[U];
        lds_pairs<CB>(lane_col + (int)(sl & 0xFF) * chunk_rows, pre);
        sl >>= 8;
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = acc[u] + pre[u];
      } break;
      case BSR_SOP_MUL_T: {
        double pre[U];
        lds_pairs<CB>(lane_col + (int)(sl & 0xFF) * chunk_rows, pre);
        sl >>= 8;
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = acc[u] * pre[u];
      } break;
      case BSR_OP_TERMINAL: {   // push
#pragma unroll
        for (int u = 0; u < U; ++u) s0[u] = acc[u];
        lds_pairs<CB>(lane_col + (int)(sl & 0xFF) * chunk_rows, acc);
        sl >>= 8;
      } break;
      case BSR_OP_ADD:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = s0[u] + acc[u];
        break;
      case BSR_OP_MUL:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = s0[u] * acc[u];
        break;
      case BSR_OP_SUB:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = s0[u] - acc[u];
        break;
      case BSR_OP_DIV:   // protected like inv
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = (acc[u] == 0.0) ? 0.0 : s0[u] / acc[u];
        break;
      case BSR_OP_INV:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = (acc[u] == 0.0) ? 0.0 : 1.0 / acc[u];
        break;
      case BSR_OP_LN: {
        const double2 ab = ln_tab[lk];   // every lane reads the same 16 bytes: a broadcast
        ++lk;
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = ab.x * acc[u] + ab.y;   // two roundings (contraction is off)
      } break;
      case BSR_OP_NEG:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = -acc[u];
        break;
      case BSR_OP_SIN:   // inline: a call would wait for the wave's LDS-DMA copies in flight (bsr_device.h: sincos_vals)
        sincos_vals<U>(acc, 0);
        break;
      case BSR_OP_COS:
        sincos_vals<U>(acc, 1);
        break;
      case BSR_OP_EXP:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = op_exp<double>(acc[u]);
        break;
      case BSR_OP_SQUARE:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = acc[u] * acc[u];
        break;
      default:  // BSR_OP_CUBIC
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = op_cube<double>(acc[u]);
        break;
    }
  }
}

// one chunk's rows of one tape into its sums (the order of accumulate_v, bsr_tile_common.h); every row lies below N
template <int KQ, int CB>
__device__ __forceinline__ void add_chunk(TapeAcc<KQ>& A, const double (&z)[2 * CB], const double (&yv)[2 * CB],
                                          const double (&qv)[KQ > 0 ? KQ : 1][2 * CB], double s, int nbc) {
  constexpr int U = 2 * CB;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (CB == 1 || u < 2 * nbc) {   // (a chunk cut short by the end of the slice: its second block is not there)
      const double zs = z[u] * s;
      A.amax = max_abs(A.amax, z[u]);
      A.a0 = fma(zs, zs, A.a0);
      A.a1 = fma(zs, yv[u], A.a1);
#pragma unroll
      for (int i = 0; i < KQ; ++i) A.c[i] = fma(qv[i][u], zs, A.c[i]);
    }
  }
}

// The block that holds row N (N not a multiple of 128) belongs to no slice: it goes out as (tape, block) units with a
// partial record of their own, rows beyond N masked, columns read through L2 (leftover_unit, bsr_tile_common.h) -- a call,
// behind the loop, when none of the wave's copies is in flight any more.
template <int KQ>
__device__ __attribute__((noinline)) void leftover_call(const TileArgs<double>* ka, int lane, int tk) {
  const uint64_t v = (uint64_t)(size_t)ka;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  using AT = TileArgs<double> CONSTANT_AS;   // (read field by field through scalar loads: no copy of the block)
  const AT& a = *(const AT*)(size_t)(((uint64_t)hi << 32) | lo);
  leftover_unit<double, KQ, AT, TileGeom CONSTANT_AS>(a, a.g, lane, tk);
}

// the sets of sums of a wave (four at K <= 4, two at K >= 5) as operands of the assembly blocks, for K = 1..8 basis columns
#define BSR_SC_SUMS_1 \
  [ca0] "+v"(A[0].c[0]), [sa0] "+v"(A[0].a0), [sb0] "+v"(A[0].a1), [am0] "+v"(A[0].amax), \
  [ca1] "+v"(A[1].c[0]), [sa1] "+v"(A[1].a0), [sb1] "+v"(A[1].a1), [am1] "+v"(A[1].amax), \
  [ca2] "+v"(A[2].c[0]), [sa2] "+v"(A[2].a0), [sb2] "+v"(A[2].a1), [am2] "+v"(A[2].amax), \
  [ca3] "+v"(A[3].c[0]), [sa3] "+v"(A[3].a0), [sb3] "+v"(A[3].a1), [am3] "+v"(A[3].amax)
#define BSR_SC_SUMS_2 \
  [ca0] "+v"(A[0].c[0]), [cb0] "+v"(A[0].c[1]), [sa0] "+v"(A[0].a0), [sb0] "+v"(A[0].a1), [am0] "+v"(A[0].amax), \
  [ca1] "+v"(A[1].c[0]), [cb1] "+v"(A[1].c[1]), [sa1] "+v"(A[1].a0), [sb1] "+v"(A[1].a1), [am1] "+v"(A[1].amax), \
  [ca2] "+v"(A[2].c[0]), [cb2] "+v"(A[2].c[1]), [sa2] "+v"(A[2].a0), [sb2] "+v"(A[2].a1), [am2] "+v"(A[2].amax), \
  [ca3] "+v"(A[3].c[0]), [cb3] "+v"(A[3].c[1]), [sa3] "+v"(A[3].a0), [sb3] "+v"(A[3].a1), [am3] "+v"(A[3].amax)
#define BSR_SC_SUMS_3 \
  [ca0] "+v"(A[0].c[0]), [cb0] "+v"(A[0].c[1]), [cc0] "+v"(A[0].c[2]), [sa0] "+v"(A[0].a0), [sb0] "+v"(A[0].a1), [am0] "+v"(A[0].amax), \
  [ca1] "+v"(A[1].c[0]), [cb1] "+v"(A[1].c[1]), [cc1] "+v"(A[1].c[2]), [sa1] "+v"(A[1].a0), [sb1] "+v"(A[1].a1), [am1] "+v"(A[1].amax), \
  [ca2] "+v"(A[2].c[0]), [cb2] "+v"(A[2].c[1]), [cc2] "+v"(A[2].c[2]), [sa2] "+v"(A[2].a0), [sb2] "+v"(A[2].a1), [am2] "+v"(A[2].amax), \
  [ca3] "+v"(A[3].c[0]), [cb3] "+v"(A[3].c[1]), [cc3] "+v"(A[3].c[2]), [sa3] "+v"(A[3].a0), [sb3] "+v"(A[3].a1), [am3] "+v"(A[3].amax)
#define BSR_SC_SUMS_4 \
  [ca0] "+v"(A[0].c[0]), [cb0] "+v"(A[0].c[1]), [cc0] "+v"(A[0].c[2]), [cd0] "+v"(A[0].c[3]), [sa0] "+v"(A[0].a0), [sb0] "+v"(A[0].a1), [am0] "+v"(A[0].amax), \
  [ca1] "+v"(A[1].c[0]), [cb1] "+v"(A[1].c[1]), [cc1] "+v"(A[1].c[2]), [cd1] "+v"(A[1].c[3]), [sa1] "+v"(A[1].a0), [sb1] "+v"(A[1].a1), [am1] "+v"(A[1].amax), \
  [ca2] "+v"(A[2].c[0]), [cb2] "+v"(A[2].c[1]), [cc2] "+v"(A[2].c[2]), [cd2] "+v"(A[2].c[3]), [sa2] "+v"(A[2].a0), [sb2] "+v"(A[2].a1), [am2] "+v"(A[2].amax), \
  [ca3] "+v"(A[3].c[0]), [cb3] "+v"(A[3].c[1]), [cc3] "+v"(A[3].c[2]), [cd3] "+v"(A[3].c[3]), [sa3] "+v"(A[3].a0), [sb3] "+v"(A[3].a1), [am3] "+v"(A[3].amax)
#define BSR_SC_SUMS_5 \
  [ca0] "+v"(A[0].c[0]), [cb0] "+v"(A[0].c[1]), [cc0] "+v"(A[0].c[2]), [cd0] "+v"(A[0].c[3]), [ce0] "+v"(A[0].c[4]), [sa0] "+v"(A[0].a0), [sb0] "+v"(A[0].a1), [am0] "+v"(A[0].amax), \
  [ca1] "+v"(A[1].c[0]), [cb1] "+v"(A[1].c[1]), [cc1] "+v"(A[1].c[2]), [cd1] "+v"(A[1].c[3]), [ce1] "+v"(A[1].c[4]), [sa1] "+v"(A[1].a0), [sb1] "+v"(A[1].a1), [am1] "+v"(A[1].amax)
#define BSR_SC_SUMS_6 \
  [ca0] "+v"(A[0].c[0]), [cb0] "+v"(A[0].c[1]), [cc0] "+v"(A[0].c[2]), [cd0] "+v"(A[0].c[3]), [ce0] "+v"(A[0].c[4]), [cf0] "+v"(A[0].c[5]), [sa0] "+v"(A[0].a0), [sb0] "+v"(A[0].a1), [am0] "+v"(A[0].amax), \
  [ca1] "+v"(A[1].c[0]), [cb1] "+v"(A[1].c[1]), [cc1] "+v"(A[1].c[2]), [cd1] "+v"(A[1].c[3]), [ce1] "+v"(A[1].c[4]), [cf1] "+v"(A[1].c[5]), [sa1] "+v"(A[1].a0), [sb1] "+v"(A[1].a1), [am1] "+v"(A[1].amax)
#define BSR_SC_SUMS_7 \
  [ca0] "+v"(A[0].c[0]), [cb0] "+v"(A[0].c[1]), [cc0] "+v"(A[0].c[2]), [cd0] "+v"(A[0].c[3]), [ce0] "+v"(A[0].c[4]), [cf0] "+v"(A[0].c[5]), [cg0] "+v"(A[0].c[6]), [sa0] "+v"(A[0].a0), [sb0] "+v"(A[0].a1), [am0] "+v"(A[0].amax), \
  [ca1] "+v"(A[1].c[0]), [cb1] "+v"(A[1].c[1]), [cc1] "+v"(A[1].c[2]), [cd1] "+v"(A[1].c[3]), [ce1] "+v"(A[1].c[4]), [cf1] "+v"(A[1].c[5]), [cg1] "+v"(A[1].c[6]), [sa1] "+v"(A[1].a0), [sb1] "+v"(A[1].a1), [am1] "+v"(A[1].amax)
#define BSR_SC_SUMS_8 \
  [ca0] "+v"(A[0].c[0]), [cb0] "+v"(A[0].c[1]), [cc0] "+v"(A[0].c[2]), [cd0] "+v"(A[0].c[3]), [ce0] "+v"(A[0].c[4]), [cf0] "+v"(A[0].c[5]), [cg0] "+v"(A[0].c[6]), [ch0] "+v"(A[0].c[7]), [sa0] "+v"(A[0].a0), [sb0] "+v"(A[0].a1), [am0] "+v"(A[0].amax), \
  [ca1] "+v"(A[1].c[0]), [cb1] "+v"(A[1].c[1]), [cc1] "+v"(A[1].c[2]), [cd1] "+v"(A[1].c[3]), [ce1] "+v"(A[1].c[4]), [cf1] "+v"(A[1].c[5]), [cg1] "+v"(A[1].c[6]), [ch1] "+v"(A[1].c[7]), [sa1] "+v"(A[1].a0), [sb1] "+v"(A[1].a1), [am1] "+v"(A[1].amax)

// A tape the fast interpreters do not take, on the lane's two rows of one block of the staged chunk.  Out of line on
// purpose: the stack machine keeps its deeper values in scratch memory, and with scratch accesses anywhere in the chunk
// loop the compiler guards every register they might still be writing with s_waitcnt vmcnt(0) -- which is also the counter
// of the wave's LDS-DMA copies in flight.  A call drains that counter too (every callee starts by waiting for everything),
// but only when such a tape comes by.
__device__ __attribute__((noinline)) double2 generic_block(const TileArgs<double>* ka, const TapeRec* recp, const double* cur,
                                                           int row0, int lane) {
  // (both pointers are the same in every lane: made scalar by hand, so that the records are read by scalar loads)
  const uint64_t v = (uint64_t)(size_t)ka, r = (uint64_t)(size_t)recp;
  const uint32_t vlo = __builtin_amdgcn_readfirstlane((uint32_t)v), vhi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  const uint32_t rlo = __builtin_amdgcn_readfirstlane((uint32_t)r), rhi = __builtin_amdgcn_readfirstlane((uint32_t)(r >> 32));
  using AT = TileArgs<double> CONSTANT_AS;
  const AT& a = *(const AT*)(size_t)(((uint64_t)vhi << 32) | vlo);
  const TapeRec CONSTANT_AS* rec = (const TapeRec CONSTANT_AS*)(size_t)(((uint64_t)rhi << 32) | rlo);
  TapeHead hd;
  hd.code0 = rec->code0; hd.code1 = rec->code1; hd.f0 = rec->f0; hd.f1 = rec->f1;
  hd.la = rec->ln[0]; hd.lb = rec->ln[1];
  hd.ln_near = (const double*)rec->ln;
  hd.n_ln = rec->n_ln;
  hd.n_term = rec->n_term;
  double zb[2];
  LdsCols<double, 2> ldr{cur, BSR_TILE_BLOCK * (int)a.g.chunk_blocks, row0};
  run_tape_head<double, 2, BSR_REG_STACK, LdsCols<double, 2>, false, true>(
      hd, a.codes + rec->code_off, a.feats + rec->feat_off, a.lnp + 2 * (size_t)rec->ln_off, rec->n_nodes, ldr, zb,
      (double*)nullptr, lane);
  return make_double2(zb[0], zb[1]);
}


// ---------------------------------------------------------------------------------------------------------------
// f32 STORAGE (round 6, BASELINE configs[4]'s fp32 context where the slices stream): the columns -- X, y, the basis, the
// derived columns -- are f32 in HBM and in the LDS ring, half the bytes of everything that moves; every value is
// converted to f64 where it is read (exact) and the interpreter, the sums and the records are the f64 ones.  A unit of
// the ring (1 KiB, one LDS-DMA instruction) is 256 rows of one column: a chunk is two 128-row blocks, evaluated one
// after the other by the chunk block of assembly (BSR_STREAM_CHUNKF_ASM_K*: ds_read_b64 + v_cvt_f64_f32 where the f64
// block has ds_read_b128).  What the C++ side of the kernel reads itself goes through these two loaders.
template <int U>
struct LdsColsF32 {
  const float* sx;
  int rb_rows;
  int off;
  __device__ __forceinline__ void load(int slot, double (&v)[U]) const {
    const float* col = sx + slot * rb_rows + off;
#pragma unroll
    for (int j = 0; j < U / 2; ++j) {
      const float2 f = *reinterpret_cast<const float2*>(col + j * 128);
      v[2 * j] = (double)f.x;
      v[2 * j + 1] = (double)f.y;
    }
  }
};
template <int U>
struct PtrColsF32 {
  const float* const CONSTANT_AS* colsrc;
  int64_t r0;
  __device__ __forceinline__ void load(int slot, double (&v)[U]) const {
    const float* col = colsrc[slot] + r0;
#pragma unroll
    for (int j = 0; j < U / 2; ++j) {
      const float2 f = *reinterpret_cast<const float2*>(col + j * 128);
      v[2 * j] = (double)f.x;
      v[2 * j + 1] = (double)f.y;
    }
  }
};
__device__ __attribute__((noinline)) double2 generic_block_f32(const TileArgs<double>* ka, const TapeRec* recp, const float* cur,
                                                               int row0, int lane) {
  const uint64_t v = (uint64_t)(size_t)ka, r = (uint64_t)(size_t)recp;
  const uint32_t vlo = __builtin_amdgcn_readfirstlane((uint32_t)v), vhi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  const uint32_t rlo = __builtin_amdgcn_readfirstlane((uint32_t)r), rhi = __builtin_amdgcn_readfirstlane((uint32_t)(r >> 32));
  using AT = TileArgs<double> CONSTANT_AS;
  const AT& a = *(const AT*)(size_t)(((uint64_t)vhi << 32) | vlo);
  const TapeRec CONSTANT_AS* rec = (const TapeRec CONSTANT_AS*)(size_t)(((uint64_t)rhi << 32) | rlo);
  TapeHead hd;
  hd.code0 = rec->code0; hd.code1 = rec->code1; hd.f0 = rec->f0; hd.f1 = rec->f1;
  hd.la = rec->ln[0]; hd.lb = rec->ln[1];
  hd.ln_near = (const double*)rec->ln;
  hd.n_ln = rec->n_ln;
  hd.n_term = rec->n_term;
  double zb[2];
  LdsColsF32<2> ldr{cur, BSR_TILE_BLOCK * (int)a.g.chunk_blocks, row0};
  run_tape_head<double, 2, BSR_REG_STACK, LdsColsF32<2>, false, true>(
      hd, a.codes + rec->code_off, a.feats + rec->feat_off, a.lnp + 2 * (size_t)rec->ln_off, rec->n_nodes, ldr, zb,
      (double*)nullptr, lane);
  return make_double2(zb[0], zb[1]);
}
// the block that holds row N (leftover_unit of bsr_tile_common.h with f32 columns read through the pointer table)
template <int KQ>
__device__ __attribute__((noinline)) void leftover_call_f32(const TileArgs<double>* ka, int lane, int tk) {
  const uint64_t v = (uint64_t)(size_t)ka;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  using AT = TileArgs<double> CONSTANT_AS;
  const AT& a = *(const AT*)(size_t)(((uint64_t)hi << 32) | lo);
  const TileGeom CONSTANT_AS& g = a.g;
  constexpr int U = BSR_TILE_U;
  const int32_t CONSTANT_AS* left_idx = as_const(reinterpret_cast<const int32_t*>(a.sched + (size_t)g.T * g.n_pass * BSR_TILE_WAVES * g.qmax));
  const int ti = tk / g.n_left, bi = tk - ti * g.n_left;
  const TapeRec CONSTANT_AS* rec = as_const(a.sched + left_idx[ti]);
  const int p = rec->p;
  const int blk = g.n_slices * g.bps + g.n_long + bi;
  const int64_t row0 = (int64_t)blk * BSR_TILE_BLOCK + 2 * lane;
  const int qslot = rec->qslot, grp = rec->grp;
  const float* const CONSTANT_AS* colsrc = (const float* const CONSTANT_AS*)group_cols<double, AT>(a, grp);
  const float2 yf = *reinterpret_cast<const float2*>(colsrc[a.grp_nF[grp & 7]] + row0);
  const double2 yv = make_double2((double)yf.x, (double)yf.y);
  double2 qv[KQ > 0 ? KQ : 1];
#pragma unroll
  for (int i = 0; i < KQ; ++i) {
    const float2 qf = *reinterpret_cast<const float2*>(colsrc[qslot + i] + row0);
    qv[i] = make_double2((double)qf.x, (double)qf.y);
  }
  TapeHead hd;
  hd.code0 = rec->code0; hd.code1 = rec->code1; hd.f0 = rec->f0; hd.f1 = rec->f1;
  hd.la = rec->ln[0]; hd.lb = rec->ln[1];
  hd.ln_near = (const double*)rec->ln;
  hd.n_ln = rec->n_ln;
  hd.n_term = rec->n_term;
  const double s = rec->s;
  double z[U];
  PtrColsF32<U> ldr{colsrc, row0};
  run_tape_head<double, U, BSR_REG_STACK, PtrColsF32<U>, false>(hd, a.codes + rec->code_off, a.feats + rec->feat_off,
                                                                a.lnp + 2 * (size_t)rec->ln_off, rec->n_nodes, ldr, z, (double*)nullptr, lane);
  TapeAcc<KQ> A;
  A.clear();
  if ((int64_t)(blk + 1) * BSR_TILE_BLOCK <= a.N) accumulate_v<double, KQ, false>(A, z, yv, qv, s, row0, a.N);
  else accumulate_v<double, KQ, true>(A, z, yv, qv, s, row0, a.N);
  store_partial<KQ>(A, a.part + ((size_t)p * g.n_part + g.n_slices + bi) * BSR_P1_WORDS, lane);
}

// (operands of the assembly blocks: the second saved value's way out of a leave and back in -- K <= 3 only)
#define BSR_S23 [s2] "+v"(s02), [s3] "+v"(s03),
#define BSR_S23_NONE
// DEEP2 (round 6; mode 3, K <= 3): the block keeps a SECOND value below the accumulator in v[40:43], for batches that hold
// trees of Strahler number 3 (a binary operator over two subtrees that each hold one).  A kernel of its own because the
// four registers cost every batch: the real mix's C5 launch measured 77 us without them and 83-86 us with (interleaved
// A/B of two builds in one box); the host picks it per batch (bsr_stage.hip) -- a tape scores the same bytes either way.
template <int KQ, int QT, int CB, bool STAMPS, int MODE, bool F32 = false, bool DEEP2 = false>
__global__ __launch_bounds__(BSR_TILE_WAVES* BSR_WAVE) void k_stream(TileArgs<double> a) {
  static_assert(!DEEP2 || (MODE == 3 && KQ <= 3), "the second saved value: the pass block at K <= 3");
  static_assert(!F32 || (CB == 2 && MODE == 2 && KQ <= 4 && QT == 4 && !STAMPS),
                "f32 storage: 256-row chunks (two blocks), the chunk block of assembly, four sets of sums per wave");
  // MODE 0: the C++ interpreter (tape_fast); 1: the assembly interpreter, a tape at a time (bsr_stream_asm.h);
  // 2: a wave's four tapes of a chunk in one block of assembly (bsr_stream_chunk_asm.h: K = 3, every tape on one basis);
  // 3: the loop over the slice's chunks inside that block too (the kernel's default where 2 applies)
  constexpr bool ASM = MODE == 1;
  static_assert(MODE != 1 || CB == 1, "the tape-at-a-time assembly interpreter takes one-block chunks");
  static_assert(MODE < 2 || (KQ <= 4 && QT == 4) || (KQ >= 5 && QT == 2),
                "the chunk block is written for four tapes per wave at K <= 4, two at K >= 5");
  static_assert(MODE != 3 || (CB == 1 && !STAMPS), "the pass block: one-block chunks, no per-chunk clock samples");
  constexpr int U = 2 * CB;
  constexpr int NUMAX = BSR_STREAM_UNITS_MAX / BSR_TILE_WAVES;   // DMA pieces per wave and chunk at most
  extern __shared__ __align__(16) unsigned char smem[];
  double* sx = reinterpret_cast<double*>(smem);   // [ring][ncols][chunk_rows], then the waves' ln pairs
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n_slices = a.g.n_slices;
  const int tg = blockIdx.x / n_slices, slice = blockIdx.x - tg * n_slices;
  const int b0 = slice * a.g.bps + min(slice, a.g.n_long);            // the first n_long slices hold one block more
  const int nb = a.g.bps + (slice < a.g.n_long ? 1 : 0);
  const int n_chunks = (nb + CB - 1) / CB;
  constexpr int chunk_rows = CB * BSR_TILE_BLOCK;
  const int R = a.g.ring;
  const int buf_elems = a.g.ncols * chunk_rows;
  const double* const CONSTANT_AS* colsrc = group_cols<double>(a, tg);
  const int y_slot = a.grp_nF[tg & 7];
  const int ncols = y_slot + 1 + a.g.ncols_fixed;
  // (f32 storage: a column of the chunk buffer holds chunk_rows f32 values)
  constexpr uint32_t ESZ = F32 ? 4u : 8u;
  double2* ln_all = reinterpret_cast<double2*>(smem + (size_t)R * buf_elems * ESZ);
  unsigned long long* stamp = STAMPS ? a.stamps + ((size_t)blockIdx.x * BSR_TILE_WAVES + wave) * BSR_TILE_STAMP_WORDS : nullptr;
#define TSTAMP(i) do { if (STAMPS && lane == 0) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
  TSTAMP(0);
  if (STAMPS && lane == 0) stamp[7] = __builtin_amdgcn_s_memrealtime();
  // the math tables into LDS by the same copies that bring the columns, requested first (copies complete in order: a
  // wave that waits for its first chunk has its piece of the tables; everyone else's after the first barrier) -- read
  // through registers they cost the kernel's start a round trip to memory before the first column was even requested
  for (int t = wave; t < (int)(BSR_TAB_DOUBLES * sizeof(double) / 1024); t += BSR_TILE_WAVES) {
    const char* src = (const char*)bsr_tables_src + t * 1024 + lane * 16;
    const uint32_t la = __builtin_amdgcn_readfirstlane(
        (uint32_t)(size_t)(__attribute__((address_space(3))) void*)((char*)bsr_lds_tab + t * 1024));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(la) : "memory", "m0");
  }
  unsigned long long busy = 0, t_busy = 0;

  // the wave's DMA pieces: piece k copies unit u = wave + 16 k of every chunk -- (column u / CB, block u % CB of the
  // chunk) -- to byte u * 1024 of the ring buffer.  Column bases are fetched from the group's table when a chunk is
  // requested (scalar loads, issued in front of the barrier the wave waits at anyway): held in scalar registers for the
  // life of the kernel they were spilled, and every reload of a spilled scalar is a vector instruction.
  const int n_units = F32 ? ncols : ncols * CB;   // (f32 storage: one 1 KiB unit is the chunk's 256 rows of a column)
  const int n_mine = max(0, (n_units - wave + BSR_TILE_WAVES - 1) / BSR_TILE_WAVES);   // the wave's pieces of a whole chunk
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)sx;
  const uint32_t buf_bytes = (uint32_t)buf_elems * ESZ, ring_bytes = buf_bytes * (uint32_t)R;
  // (two-block chunks: u % 2 == wave % 2 for every piece of the wave; the last chunk of a slice of an odd number of
  // blocks holds block 0 only: the odd waves then copy nothing)
  // (f32 storage: a unit always holds both blocks; the second half of a slice's last, odd chunk is the next slice's first
  // block -- or the column's padding -- copied and never evaluated)
  auto pieces = [&](int j) { return (!F32 && CB == 2 && (j + 1) * CB > nb && (wave & 1)) ? 0 : n_mine; };
  struct Bases { uint64_t p[NUMAX]; };
  auto fetch_bases = [&]() {
    Bases B;
#pragma unroll
    for (int k = 0; k < NUMAX; ++k) {
      int u = min(wave + BSR_TILE_WAVES * k, n_units - 1);   // (a piece the wave does not have reads a valid entry)
      asm volatile("" : "+s"(u));   // (not loop-invariant to the compiler: hoisted, the bases would live -- spilled -- in
                                    // scalar registers for the whole kernel again)
      B.p[k] = (uint64_t)(size_t)colsrc[F32 ? u : u / CB];
    }
    return B;
  };
  // chunk j into the ring buffer at byte `boff` of the ring
  auto issue = [&](int j, uint32_t boff, const Bases& B) {
    int np = pieces(j);
    asm volatile("" : "+s"(np));   // (compared where it is used: precomputed, the four conditions are four spilled lane masks)
    const uint32_t buf = lds0 + boff + (uint32_t)wave * 1024u;
    const uint32_t voff = F32 ? (uint32_t)lane * 16u + (uint32_t)(b0 + j * CB) * 512u
                              : (uint32_t)lane * 16u + (uint32_t)(b0 + j * CB + (CB == 2 ? (wave & 1) : 0)) * 1024u;
#pragma unroll
    for (int k = 0; k < NUMAX; ++k) {
      if (k < np) {
        const uint32_t la = buf + (uint32_t)k * (BSR_TILE_WAVES * 1024u);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(B.p[k]), "s"(la) : "memory", "m0");
      }
    }
  };
  const int n_fixed = a.g.ncols_fixed;   // == KQ: every tape of the batch projects on the same basis (slots y_slot + 1 ..)

  for (int pass = 0; pass < a.g.n_pass; ++pass) {
    const TapeRec* my = a.sched + (((size_t)tg * a.g.n_pass + pass) * BSR_TILE_WAVES + wave) * QT;
    // the wave's tapes: ln pairs into LDS, sums cleared; their programs (StreamRec, 32 bytes) are read again for every
    // chunk by one scalar load each, requested under the sums of the tape before
    TapeAcc<KQ> A[QT];
    double2* ln_mine = ln_all + (size_t)wave * QT * BSR_STREAM_LN_PAIRS;
    const uint32_t ln_lds = lds0 + ring_bytes + (uint32_t)wave * (QT * BSR_STREAM_LN_PAIRS * 16u);   // (its LDS address, for the assembly interpreter)
    if (pass != 0) __syncthreads();   // everyone is done with the last chunks (and the ln pairs) of the pass before
    const StreamRec CONSTANT_AS* sr = as_const(a.srec + (((size_t)tg * a.g.n_pass + pass) * BSR_TILE_WAVES + wave) * QT);
    uint32_t issue_off = 0;   // where in the ring the next chunk to be requested goes
    {
      const Bases B = fetch_bases();
      for (int j = 0; j < min(R - 1, n_chunks); ++j) {
        issue(j, issue_off, B);
        issue_off += buf_bytes;
      }
    }
    if (issue_off == ring_bytes) issue_off = 0;
#pragma unroll
    for (int q = 0; q < QT; ++q) {   // (under the first copies' flight)
      const TapeRec CONSTANT_AS* rec = as_const(my + q);
      A[q].clear();
      if (lane == 0) {   // (scalar loads and LDS stores: no vector memory operation next to the copies' counter)
#pragma unroll
        for (int j = 0; j < 3; ++j) ln_mine[q * BSR_STREAM_LN_PAIRS + j] = make_double2(rec->ln[2 * j], rec->ln[2 * j + 1]);
        // (round 6: a long tape's pairs 4..8 from its ln stream -- the record holds the first three)
        const int nl = rec->n_ln;
        if (rec->p >= 0 && nl > 2) {   // (a set of sums without a tape: its record holds nothing but p = -1)
          const double CONSTANT_AS* lp = as_const(a.lnp + 2 * (size_t)rec->ln_off);
          for (int j = 3; j < BSR_STREAM_LN_PAIRS && j <= nl; ++j) ln_mine[q * BSR_STREAM_LN_PAIRS + j] = make_double2(lp[2 * j], lp[2 * j + 1]);
        }
      }
    }
    uint32_t cur_off = 0;     // ... and where the chunk the waves compute on sits
    if constexpr (MODE == 3) {
      // the chunk loop itself in the block of assembly (bsr_stream_chunk_asm.h: BSR_STREAM_PASS_ASM_K3); what comes back
      // here: sin / cos of huge arguments and tapes for the stack machine, with the loop's state in sv[5..9]
      const Bases B = fetch_bases();
      uint32_t resume = 0, st, sv[10], lc_out;
      double z0, z1, s00, s01, s02, s03;   // (s02, s03: the second value below the accumulator, K <= 3)
      asm volatile("" : "=v"(z0), "=v"(z1), "=v"(s00), "=v"(s01), "=v"(s02), "=v"(s03), "=v"(sv[0]), "=v"(sv[1]), "=v"(sv[2]), "=v"(sv[3]),
                        "=v"(sv[4]), "=v"(sv[5]), "=v"(sv[6]), "=v"(sv[7]), "=v"(sv[8]), "=v"(sv[9]));
      const uint32_t yo = (uint32_t)y_slot << 10, lane16 = (uint32_t)lane * 16u;
      const uint32_t tab_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)bsr_lds_tab;
      const uint32_t ring_u = (uint32_t)R, nch_u = (uint32_t)n_chunks, nmine_u = (uint32_t)n_mine, wave_u = (uint32_t)wave;
      const uint32_t b0_u = (uint32_t)b0;
      for (;;) {
#define BSR_SP_OPERANDS(SUMS, S23) \
                     : SUMS, \
                       [z0] "+v"(z0), [z1] "+v"(z1), [s0] "+v"(s00), [s1] "+v"(s01), S23 [sv0] "+v"(sv[0]), \
                       [sv1] "+v"(sv[1]), [sv2] "+v"(sv[2]), [sv3] "+v"(sv[3]), [sv4] "+v"(sv[4]), [sv5] "+v"(sv[5]), \
                       [sv6] "+v"(sv[6]), [sv7] "+v"(sv[7]), [sv8] "+v"(sv[8]), [sv9] "+v"(sv[9]), [st] "=s"(st), \
                       [lc] "=&v"(lc_out) \
                     : [resume] "s"(resume), [sr] "s"(sr), [ln] "s"(ln_lds), [yo] "s"(yo), [tab] "s"(tab_lds), \
                       [lane16] "v"(lane16), [ba0] "s"(B.p[0]), [ba1] "s"(B.p[1]), [ba2] "s"(B.p[2]), [ba3] "s"(B.p[3]), \
                       [nch] "s"(nch_u), [bufb] "s"(buf_bytes), [ring] "s"(ring_u), [lds0] "s"(lds0), [wave] "s"(wave_u), \
                       [nmine] "s"(nmine_u), [ioff] "s"(issue_off), [b0] "s"(b0_u)
        if constexpr (DEEP2 && KQ == 1) asm volatile(BSR_STREAM_PASS_ASM_K1D BSR_SP_OPERANDS(BSR_SC_SUMS_1, BSR_S23) : BSR_STREAM_PASS_CLOBBERS_K4);
        else if constexpr (DEEP2 && KQ == 2) asm volatile(BSR_STREAM_PASS_ASM_K2D BSR_SP_OPERANDS(BSR_SC_SUMS_2, BSR_S23) : BSR_STREAM_PASS_CLOBBERS_K4);
        else if constexpr (DEEP2 && KQ == 3) asm volatile(BSR_STREAM_PASS_ASM_K3D BSR_SP_OPERANDS(BSR_SC_SUMS_3, BSR_S23) : BSR_STREAM_PASS_CLOBBERS_K4);
        else if constexpr (KQ == 1) asm volatile(BSR_STREAM_PASS_ASM_K1 BSR_SP_OPERANDS(BSR_SC_SUMS_1, BSR_S23_NONE) : BSR_STREAM_PASS_CLOBBERS);
        else if constexpr (KQ == 2) asm volatile(BSR_STREAM_PASS_ASM_K2 BSR_SP_OPERANDS(BSR_SC_SUMS_2, BSR_S23_NONE) : BSR_STREAM_PASS_CLOBBERS);
        else if constexpr (KQ == 3) asm volatile(BSR_STREAM_PASS_ASM_K3 BSR_SP_OPERANDS(BSR_SC_SUMS_3, BSR_S23_NONE) : BSR_STREAM_PASS_CLOBBERS);
        else if constexpr (KQ == 4) asm volatile(BSR_STREAM_PASS_ASM_K4 BSR_SP_OPERANDS(BSR_SC_SUMS_4, BSR_S23_NONE) : BSR_STREAM_PASS_CLOBBERS_K4);
        else if constexpr (KQ == 5) asm volatile(BSR_STREAM_PASS_ASM_K5 BSR_SP_OPERANDS(BSR_SC_SUMS_5, BSR_S23_NONE) : BSR_STREAM_PASS_CLOBBERS_K8);
        else if constexpr (KQ == 6) asm volatile(BSR_STREAM_PASS_ASM_K6 BSR_SP_OPERANDS(BSR_SC_SUMS_6, BSR_S23_NONE) : BSR_STREAM_PASS_CLOBBERS_K8);
        else if constexpr (KQ == 7) asm volatile(BSR_STREAM_PASS_ASM_K7 BSR_SP_OPERANDS(BSR_SC_SUMS_7, BSR_S23_NONE) : BSR_STREAM_PASS_CLOBBERS_K8);
        else asm volatile(BSR_STREAM_PASS_ASM_K8 BSR_SP_OPERANDS(BSR_SC_SUMS_8, BSR_S23_NONE) : BSR_STREAM_PASS_CLOBBERS_K8);
#undef BSR_SP_OPERANDS
        if (st == 0) break;
        const uint32_t what = st & 15u;
        if (what == 1) {
          const uint32_t cur_lds = __builtin_amdgcn_readfirstlane(sv[9]);   // LDS address of the chunk being computed on
          const double2 zb = generic_block((const TileArgs<double>*)__builtin_amdgcn_kernarg_segment_ptr(), my + (st >> 4),
                                           reinterpret_cast<const double*>(smem + (cur_lds - lds0)), 2 * lane, lane);
          z0 = zb.x;
          z1 = zb.y;
        } else {
          double zz[2] = {z0, z1};
          sincos_vals<2>(zz, what == BSR_OP_COS + 1 ? 1 : 0);
          z0 = zz[0];
          z1 = zz[1];
        }
        resume = __builtin_amdgcn_readfirstlane(st);
      }
    } else
    for (int ci = 0; ci < n_chunks; ++ci) {
      const Bases B = fetch_bases();   // (for the request behind the barrier; the loads return while the wave waits there)
      {
        // copies complete in order: chunk ci has landed when at most the pieces of the chunks requested behind it are
        // still in flight
        int later;
        if (CB == 1) {
          later = n_mine * min(R - 2, n_chunks - 1 - ci);
        } else {
          later = 0;
          for (int j = ci + 1; j < min(ci + R - 1, n_chunks); ++j) later += pieces(j);
        }
        dma_wait_left(later);
      }
      __syncthreads();   // chunk ci has landed for everyone; everyone is done with chunk ci - 1
      if (ci + R - 1 < n_chunks) {
        issue(ci + R - 1, issue_off, B);
        issue_off += buf_bytes;
        if (issue_off == ring_bytes) issue_off = 0;
      }
      if (ci == 0 && pass == 0) TSTAMP(1);
      if (STAMPS) t_busy = __builtin_amdgcn_s_memtime();
      const double* cur = reinterpret_cast<const double*>(smem + cur_off);
      const uint32_t lc = lds0 + cur_off + (uint32_t)lane * (F32 ? 8u : 16u);   // LDS address of the lane's pair in column 0
      cur_off += buf_bytes;
      if (cur_off == ring_bytes) cur_off = 0;
      const double* lane_col = cur + 2 * lane;
      const int nbc = (CB == 1) ? 1 : min(CB, nb - ci * CB);
      // One-block chunks, K <= 3: y (and below: the basis columns) of the lane's rows are read once for all tapes of the
      // wave.  Two-block chunks, or more basis columns: every tape reads them again when its values are ready -- held across the tapes they are 32
      // registers, which with four sets of sums and the evaluation's temporaries is more than a wave has.
      // (the assembly interpreter's fixed registers leave no room for them either)
      constexpr bool HOLD = CB == 1 && KQ <= 3 && !ASM;   // (K >= 4: the basis values alone are 16+ registers per block)
      double yv[U];
      double qv[KQ > 0 ? KQ : 1][U];
      int q_have = -1;
      int nfx = n_fixed;
      asm volatile("" : "+s"(nfx));
      if (HOLD) {
        lds_pairs<CB>(lane_col + y_slot * chunk_rows, yv);
        if (nfx == KQ) {
          q_have = y_slot + 1;
#pragma unroll
          for (int i = 0; i < KQ; ++i) lds_pairs<CB>(lane_col + (y_slot + 1 + i) * chunk_rows, qv[i]);
        }
      }
      if constexpr (MODE == 2) {
        // the four tapes of the wave in one block of assembly; sin, cos, exp and tapes for the stack machine come back
        // here, and the block is entered again where it left (`resume`)
        uint32_t resume = 0, st, sv[5];
        double z0, z1, s00, s01, s02, s03;
        asm volatile("" : "=v"(z0), "=v"(z1), "=v"(s00), "=v"(s01), "=v"(s02), "=v"(s03), "=v"(sv[0]), "=v"(sv[1]), "=v"(sv[2]), "=v"(sv[3]),
                          "=v"(sv[4]));   // (no value yet: nothing to initialise)
        const uint32_t yo = (uint32_t)y_slot << ((CB == 2 && !F32) ? 11 : 10);
        const uint32_t tab_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)bsr_lds_tab;
        // (two-block chunks: the same block of assembly on either half -- a lane's rows reach its sums in the order of
        // one-block chunks, block by block: the same sums bit for bit, with half the barriers)
#define BSR_SC_OPERANDS(SUMS, S23)                                                                                     \
  : SUMS, [z0] "+v"(z0), [z1] "+v"(z1), [s0] "+v"(s00), [s1] "+v"(s01), S23 [sv0] "+v"(sv[0]), [sv1] "+v"(sv[1]),            \
    [sv2] "+v"(sv[2]), [sv3] "+v"(sv[3]), [sv4] "+v"(sv[4]), [st] "=s"(st)                                              \
  : [resume] "s"(resume), [lc] "v"(lcb), [sr] "s"(sr), [ln] "s"(ln_lds), [yo] "s"(yo), [tab] "s"(tab_lds)                \
  :
#define BSR_SC_EMIT(P)                                                                                                 \
  if constexpr (KQ == 1) asm volatile(P##1 BSR_SC_OPERANDS(BSR_SC_SUMS_1, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS);                \
  else if constexpr (KQ == 2) asm volatile(P##2 BSR_SC_OPERANDS(BSR_SC_SUMS_2, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS);           \
  else if constexpr (KQ == 3) asm volatile(P##3 BSR_SC_OPERANDS(BSR_SC_SUMS_3, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS);           \
  else if constexpr (KQ == 4) asm volatile(P##4 BSR_SC_OPERANDS(BSR_SC_SUMS_4, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS_K4);           \
  else if constexpr (KQ == 5) asm volatile(P##5 BSR_SC_OPERANDS(BSR_SC_SUMS_5, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS_K8);           \
  else if constexpr (KQ == 6) asm volatile(P##6 BSR_SC_OPERANDS(BSR_SC_SUMS_6, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS_K8);           \
  else if constexpr (KQ == 7) asm volatile(P##7 BSR_SC_OPERANDS(BSR_SC_SUMS_7, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS_K8);           \
  else asm volatile(P##8 BSR_SC_OPERANDS(BSR_SC_SUMS_8, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS_K8)
#pragma unroll 1
        for (int jb = 0; jb < nbc; ++jb) {
          const uint32_t lcb = lc + (uint32_t)jb * (F32 ? 512u : 1024u);
          resume = 0;
          for (;;) {
            if constexpr (F32) {
              if constexpr (KQ == 1) asm volatile(BSR_STREAM_CHUNKF_ASM_K1 BSR_SC_OPERANDS(BSR_SC_SUMS_1, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS);
              else if constexpr (KQ == 2) asm volatile(BSR_STREAM_CHUNKF_ASM_K2 BSR_SC_OPERANDS(BSR_SC_SUMS_2, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS);
              else if constexpr (KQ == 3) asm volatile(BSR_STREAM_CHUNKF_ASM_K3 BSR_SC_OPERANDS(BSR_SC_SUMS_3, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS);
              else asm volatile(BSR_STREAM_CHUNKF_ASM_K4 BSR_SC_OPERANDS(BSR_SC_SUMS_4, BSR_S23_NONE) BSR_STREAM_CHUNK_CLOBBERS_K4);
            } else if constexpr (CB == 2) {
              BSR_SC_EMIT(BSR_STREAM_CHUNK2_ASM_K);
            } else {
              BSR_SC_EMIT(BSR_STREAM_CHUNK_ASM_K);
            }
            if (st == 0) break;
            const uint32_t what = st & 15u;
            if (what == 1) {
              double2 zb;
              if constexpr (F32)
                zb = generic_block_f32((const TileArgs<double>*)__builtin_amdgcn_kernarg_segment_ptr(), my + (st >> 4),
                                       reinterpret_cast<const float*>(cur), jb * BSR_TILE_BLOCK + 2 * lane, lane);
              else
                zb = generic_block((const TileArgs<double>*)__builtin_amdgcn_kernarg_segment_ptr(),
                                   my + (st >> 4), cur, jb * BSR_TILE_BLOCK + 2 * lane, lane);
              z0 = zb.x;
              z1 = zb.y;
            } else {   // sin / cos of huge arguments (exp never leaves the block)
              double zz[2] = {z0, z1};
              sincos_vals<2>(zz, what == BSR_OP_COS + 1 ? 1 : 0);
              z0 = zz[0];
              z1 = zz[1];
            }
            resume = __builtin_amdgcn_readfirstlane(st);
          }
        }
#undef BSR_SC_EMIT
#undef BSR_SC_OPERANDS
        if (STAMPS) busy += __builtin_amdgcn_s_memtime() - t_busy;
        if (ci == 0 && pass == 0) TSTAMP(2);
        continue;
      }
      TapeS nx = load_tape(sr);
#pragma unroll 1
      for (int q = 0; q < QT; ++q) {
        const TapeS t = nx;
        nx = load_tape(sr + q + 1);   // the next tape's program: its scalar load returns under this tape's first LDS wait
                                      // (one record of padding behind the last).  (Tried: the programs in one vector
                                      // register, read by v_readlane -- no memory round trip, eight vector instructions
                                      // per tape: 108 -> 132 us.  Vector issue is what this kernel has least of.)
        if (!t.any()) continue;
        double z[U];
        if (!(ASM && t.fast())) lds_pairs<CB>(lane_col + (int)(t.first >> (CB == 2 ? 11 : 10)) * chunk_rows, z);
        if (HOLD && nfx != KQ && t.qslot() != q_have) {   // (tapes of one chain share the basis: read once)
          q_have = t.qslot();
#pragma unroll
          for (int i = 0; i < KQ; ++i) lds_pairs<CB>(lane_col + (q_have + i) * chunk_rows, qv[i]);
        }
        if (ASM && __builtin_expect(t.fast(), 1)) {
          if constexpr (ASM) {
            // the interpreter of bsr_stream_asm.h; sin, cos and exp come back here with the state in the operands
            // (the state of a tape that left for sin / cos / exp travels in five vector registers: tied scalar operands
            // carried around this loop do not compile -- "illegal VGPR to SGPR copy")
            const uint64_t code = t.code, sl = t.slots;
            const uint32_t lnp = ln_lds + (uint32_t)q * (BSR_STREAM_LN_PAIRS * 16u), first = t.first;
            uint32_t resume = 0, st, sv[5];
            double s00, s01;
            asm volatile("" : "=v"(z[0]), "=v"(z[1]), "=v"(s00), "=v"(s01), "=v"(sv[0]), "=v"(sv[1]), "=v"(sv[2]),
                              "=v"(sv[3]), "=v"(sv[4]));   // (no value yet: nothing to initialise)
            for (;;) {
              asm volatile(BSR_STREAM_INTERP_ASM
                           : [z0] "+v"(z[0]), [z1] "+v"(z[1]), [s0] "+v"(s00), [s1] "+v"(s01), [sv0] "+v"(sv[0]),
                             [sv1] "+v"(sv[1]), [sv2] "+v"(sv[2]), [sv3] "+v"(sv[3]), [sv4] "+v"(sv[4]), [st] "=s"(st)
                           : [resume] "s"(resume), [first] "s"(first), [lc] "v"(lc), [code] "s"(code), [sl] "s"(sl),
                             [ln] "s"(lnp)
                           : BSR_STREAM_INTERP_CLOBBERS);
              if (st == 0) break;
              if (st == BSR_OP_EXP + 1) {
                z[0] = op_exp<double>(z[0]);
                z[1] = op_exp<double>(z[1]);
              } else {
                sincos_vals<2>(reinterpret_cast<double(&)[2]>(z), st == BSR_OP_COS + 1 ? 1 : 0);
              }
              resume = 1;
            }
          }
        } else if (t.fast()) {
          tape_fast<CB>(t, lane_col, chunk_rows, ln_mine + q * BSR_STREAM_LN_PAIRS, z);
        } else {
          // Any other tape (not a chain; longer than the scalar registers hold, or with a `log`): the stack machine of
          // bsr_device.h on the same rows, one block at a time, out of line (generic_block above)
#pragma unroll 1
          for (int jb = 0; jb < CB; ++jb) {
            const double2 zb = generic_block((const TileArgs<double>*)__builtin_amdgcn_kernarg_segment_ptr(), my + q, cur,
                                             jb * BSR_TILE_BLOCK + 2 * lane, lane);
            if (CB == 1 || jb == 0) { z[0] = zb.x; z[1] = zb.y; }
            else { z[U - 2] = zb.x; z[U - 1] = zb.y; }
          }
        }
        if (!HOLD) {
          lds_pairs<CB>(lane_col + y_slot * chunk_rows, yv);
          const int qs = t.qslot();
#pragma unroll
          for (int i = 0; i < KQ; ++i) lds_pairs<CB>(lane_col + (qs + i) * chunk_rows, qv[i]);
        }
        switch (q) {
#define BSR_ADD_CASE(qq) case qq: if constexpr (qq < QT) add_chunk<KQ, CB>(A[qq], z, yv, qv, t.s, nbc); break;
          BSR_ADD_CASE(0) BSR_ADD_CASE(1) BSR_ADD_CASE(2) BSR_ADD_CASE(3)
#undef BSR_ADD_CASE
        }
      }
      if (STAMPS) busy += __builtin_amdgcn_s_memtime() - t_busy;
      if (ci == 0 && pass == 0) TSTAMP(2);
    }
    if (pass == a.g.n_pass - 1) TSTAMP(3);
    if (STAMPS && lane == 0) stamp[5] = busy;
    __syncthreads();   // everyone is done with the ring: its first bytes serve as the reductions' scratch
    reduce_store<KQ, QT>(A, my, a.part, a.g.n_part, slice, sx + (size_t)wave * (QT * (KQ + 2) + 12), lane);
  }
  // the block that holds row N, if N is not a multiple of 128: unit u of the launch goes to wave u mod (workgroups x 16)
  if (a.g.n_left > 0) {
    const int n_units_left = a.P * a.g.n_left, n_waves = (int)gridDim.x * BSR_TILE_WAVES;
    for (int tk = wave * (int)gridDim.x + (int)blockIdx.x; tk < n_units_left; tk += n_waves) {
      if constexpr (F32) leftover_call_f32<KQ>((const TileArgs<double>*)__builtin_amdgcn_kernarg_segment_ptr(), lane, tk);
      else leftover_call<KQ>((const TileArgs<double>*)__builtin_amdgcn_kernarg_segment_ptr(), lane, tk);
    }
  }
  TSTAMP(4);
  if (STAMPS && lane == 0) stamp[6] = __builtin_amdgcn_s_memrealtime();
#undef TSTAMP
}

template <int KQ, int QT, int CB, bool STAMPS, int MODE, bool F32 = false, bool DEEP2 = false>
void launch_one(hipStream_t st, const TileArgs<double>& a, size_t lds) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_stream<KQ, QT, CB, STAMPS, MODE, F32, DEEP2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(tile_lds_bytes_max() - 1024));
    attr = true;
  }
  const dim3 grid((unsigned)(a.g.T * a.g.n_slices)), block(BSR_TILE_WAVES * BSR_WAVE);
  bsr_launch((k_stream<KQ, QT, CB, STAMPS, MODE, F32, DEEP2>), grid, block, lds, st, a);
}
#ifdef BSR_TEST_VARIANTS
// (the test build: every interpreter behind BSR_STREAM_ASM, per-wave clock samples, either number of sets of sums)
template <int KQ, int QT>
void launch_cb(hipStream_t st, const TileArgs<double>& a, size_t lds, bool deep2) {
  // BSR_STREAM_ASM=0: the C++ interpreter (tape_fast) on one-block chunks too -- the cross-check of the assembly ones;
  // 1: the assembly interpreter a tape at a time; default: the wave's four tapes in one block where it applies (K = 3,
  // one chain's basis behind y), else 1
  static const int asm_mode = env_int("BSR_STREAM_ASM", 3);
  constexpr bool BLOCK_SHAPE = (KQ <= 4 && QT == 4) || (KQ >= 5 && QT == 2);
  const bool chunk_block = BLOCK_SHAPE && asm_mode >= 2 && a.g.ncols_fixed == KQ;
  if (a.g.chunk_blocks == 2) {
    if constexpr (BLOCK_SHAPE) {
      if (chunk_block) {
        if (a.stamps) launch_one<KQ, QT, 2, true, 2>(st, a, lds);
        else launch_one<KQ, QT, 2, false, 2>(st, a, lds);
        return;
      }
    }
    if (a.stamps) launch_one<KQ, QT, 2, true, 0>(st, a, lds);
    else launch_one<KQ, QT, 2, false, 0>(st, a, lds);
    return;
  }
  if constexpr (BLOCK_SHAPE) {
    if (chunk_block) {
      if (a.stamps) launch_one<KQ, QT, 1, true, 2>(st, a, lds);   // (BSR_TILE_STAMPS=1: per-wave clock samples)
      else if (asm_mode >= 3) {
        if constexpr (KQ <= 3) {
          if (deep2) { launch_one<KQ, QT, 1, false, 3, false, true>(st, a, lds); return; }
        }
        launch_one<KQ, QT, 1, false, 3>(st, a, lds);
      }
      else launch_one<KQ, QT, 1, false, 2>(st, a, lds);
      return;
    }
  }
  if (a.stamps) launch_one<KQ, QT, 1, true, 1>(st, a, lds);
  else if (asm_mode >= 1) launch_one<KQ, QT, 1, false, 1>(st, a, lds);
  else launch_one<KQ, QT, 1, false, 0>(st, a, lds);
}
#else
// The shipped library holds ONE kernel per situation: a batch on one chain's basis takes the block of assembly -- the whole
// pass in it on one-block chunks (mode 3), a chunk at a time on two-block chunks (mode 2); a batch that spans several
// chains' bases the assembly interpreter a tape at a time (mode 1; two-block chunks: the C++ interpreter, mode 0).  The
// other interpreters exist for the byte-equality tests only (build with BSR_EXTRA_FLAGS=-DBSR_TEST_VARIANTS:
// csrc/build.sh variants; tests/test_gpu_stream.py runs them).
template <int KQ, int QT>
void launch_cb(hipStream_t st, const TileArgs<double>& a, size_t lds, bool deep2) {
  static_assert((KQ <= 4 && QT == 4) || (KQ >= 5 && QT == 2), "the sets of sums per wave the kernel is built for");
  const bool chunk_block = a.g.ncols_fixed == KQ;
  if (a.g.chunk_blocks == 2) {
    if (chunk_block) launch_one<KQ, QT, 2, false, 2>(st, a, lds);
    else launch_one<KQ, QT, 2, false, 0>(st, a, lds);
    return;
  }
  if constexpr (KQ <= 3) {
    if (chunk_block && deep2) { launch_one<KQ, QT, 1, false, 3, false, true>(st, a, lds); return; }
  }
  if (chunk_block) launch_one<KQ, QT, 1, false, 3>(st, a, lds);
  else launch_one<KQ, QT, 1, false, 1>(st, a, lds);
}
#endif

}  // namespace

// bytes of LDS behind the ring: BSR_STREAM_LN_PAIRS (a, b) pairs per tape of every wave
size_t stream_ln_bytes(int qt) { return (size_t)BSR_TILE_WAVES * qt * BSR_STREAM_LN_PAIRS * sizeof(double2); }

// sets of sums per wave of the streaming kernel (BSR_STREAM_QT overrides: 2 or 4): four while the sums of four tapes, a
// chunk's y and basis values and the routines' temporaries fit 128 registers (K <= 4), else two
// whether a batch of this shape takes the chunk block of assembly (K = 3, four sets of sums per wave, one chain's basis):
// then two-block chunks cost nothing but LDS (bsr_stage.hip: stage_tile's geometry)
bool stream_chunk_block(int K, int ncols_fixed) {
#ifdef BSR_TEST_VARIANTS
  static const int asm_mode = env_int("BSR_STREAM_ASM", 3);
#else
  const int asm_mode = 3;
#endif
  return stream_qmax(K) == (K <= 4 ? 4 : 2) && asm_mode >= 2 && ncols_fixed == K;
}

// whether a batch of this shape may use the kernel with the second saved value (mode 3: one-block chunks, one chain's basis, K <= 3)
bool stream_deep2_applies(int K, int ncols_fixed, int chunk_blocks) {
#ifdef BSR_TEST_VARIANTS
  static const int asm_mode = env_int("BSR_STREAM_ASM", 3);
  if (asm_mode < 3) return false;
#endif
  return K <= 3 && ncols_fixed == K && chunk_blocks == 1 && stream_qmax(K) == 4;
}

int stream_qmax(int K) {
#ifdef BSR_TEST_VARIANTS
  static const int forced = env_int("BSR_STREAM_QT", 0);
  if (forced == 2 || (forced == 4 && K <= 4)) return forced;
#endif
  return K <= 4 ? 4 : 2;
}

// f32 storage (K <= 4; TileArgs<float> has TileArgs<double>'s layout: the column pointers are addresses to the kernel)
void launch_stream_f32(hipStream_t st, const TileArgs<float>& af) {
  static_assert(sizeof(TileArgs<float>) == sizeof(TileArgs<double>), "one argument block");
  const TileArgs<double>& a = reinterpret_cast<const TileArgs<double>&>(af);
  const TileGeom& g = a.g;
  const size_t lds = (size_t)g.ring * g.ncols * 1024 + stream_ln_bytes(g.qmax);
  switch (a.K) {
    case 1: launch_one<1, 4, 2, false, 2, true>(st, a, lds); break;
    case 2: launch_one<2, 4, 2, false, 2, true>(st, a, lds); break;
    case 3: launch_one<3, 4, 2, false, 2, true>(st, a, lds); break;
    default: launch_one<4, 4, 2, false, 2, true>(st, a, lds); break;
  }
}

void launch_stream(hipStream_t st, const TileArgs<double>& a, bool deep2) {
  const TileGeom& g = a.g;
  const size_t lds = (size_t)g.ring * g.ncols * g.chunk_blocks * BSR_TILE_BLOCK * sizeof(double) + stream_ln_bytes(g.qmax);
  if (g.qmax == 4) {
    switch (a.K) {
      case 1: launch_cb<1, 4>(st, a, lds, deep2); break;
      case 2: launch_cb<2, 4>(st, a, lds, deep2); break;
      case 3: launch_cb<3, 4>(st, a, lds, deep2); break;
      default: launch_cb<4, 4>(st, a, lds, deep2); break;
    }
    return;
  }
  switch (a.K) {
#ifdef BSR_TEST_VARIANTS
    case 1: launch_cb<1, 2>(st, a, lds, deep2); break;
    case 2: launch_cb<2, 2>(st, a, lds, deep2); break;
    case 3: launch_cb<3, 2>(st, a, lds, deep2); break;
    case 4: launch_cb<4, 2>(st, a, lds, deep2); break;
#endif
    case 5: launch_cb<5, 2>(st, a, lds, deep2); break;
    case 6: launch_cb<6, 2>(st, a, lds, deep2); break;
    case 7: launch_cb<7, 2>(st, a, lds, deep2); break;
    default: launch_cb<8, 2>(st, a, lds, deep2); break;
  }
}
