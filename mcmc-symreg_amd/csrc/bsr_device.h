// Device-side building blocks shared by the row-pass kernels (bsr_kernels.hip: k_rows, bsr_tile.hip: k_tile):
// DPP wave reductions, the opcode semantics of codes/funcs.py:179-212, the register value stack and the postfix
// stack-machine interpreter over the compact tape streams.  gfx950 only (wave64).
#pragma once

#include "bsr_internal.h"
#define BSR_TABLE_QUALIFIER static __device__ const
#include "bsr_tables.h"
#include "bsr_fastmath.h"

// The sin/cos/exp tables (9 KiB) live in LDS: every kernel that runs the interpreter copies them in once
// (tables_to_lds + a workgroup barrier) before its first tape.
static __shared__ double bsr_lds_tab[BSR_TAB_DOUBLES];
__device__ __forceinline__ void tables_to_lds() {
  for (int i = threadIdx.x; i < BSR_TAB_DOUBLES; i += blockDim.x) bsr_lds_tab[i] = bsr_tables_src[i];
}

#define CONSTANT_AS __attribute__((address_space(4)))

template <typename T>
__device__ __forceinline__ const T CONSTANT_AS* as_const(const T* p) {
  return (const T CONSTANT_AS*)p;
}

// Wave-level reductions on the VALU with DPP (no LDS traffic, unlike __shfl which lowers to ds_bpermute).
// Fixed combination order -> deterministic.  After the six steps lane 63 holds the reduction of all 64 lanes;
// v_readlane broadcasts it.  DPP controls: quad_perm 0xB1 = [1,0,3,2], 0x4E = [2,3,0,1], 0x141 row_half_mirror,
// 0x140 row_mirror, 0x142 row_bcast:15 (rows 1,3), 0x143 row_bcast:31 (rows 2,3).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int l2, h2;
  if constexpr (ROW_MASK == 0xF) {  // every lane is written: no need to initialise the destination
    l2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    h2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
  } else {
    l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  }
  return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double readlane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
// max(a, |b|) in one instruction.  fmax() costs three here: LLVM canonicalises both operands (v_max_f64 x, x) before
// the real maximum to quiet signalling NaNs.  v_max_f64 already returns the other operand when one is a NaN, which is
// the fmax behaviour the census relies on (a NaN row never becomes the maximum; it is flagged through the sum).
__device__ __forceinline__ double max_abs(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max_abs(float a, float b) { return fmaxf(a, fabsf(b)); }
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1, 0xF>(v);
  v += dpp_f64<0x4E, 0xF>(v);
  v += dpp_f64<0x141, 0xF>(v);
  v += dpp_f64<0x140, 0xF>(v);
  // rows not selected by the row mask receive 0 (old = 0, bound_ctrl off keeps `old`): add is a no-op there
  v += dpp_f64<0x142, 0xA>(v);
  v += dpp_f64<0x143, 0xC>(v);
  return readlane63(v);
}
__device__ __forceinline__ double wave_max(double v) {  // inputs are non-negative (|z| maxima): 0 is neutral
  // plain fmax here: its first step also re-materialises v through an ordinary VALU instruction, so the DPP reads below
  // never follow the inline-asm write of max_abs directly (the hazard recogniser cannot see into inline asm)
  v = fmax(v, 0.0);
  v = fmax(v, dpp_f64<0xB1, 0xF>(v));
  v = fmax(v, dpp_f64<0x4E, 0xF>(v));
  v = fmax(v, dpp_f64<0x141, 0xF>(v));
  v = fmax(v, dpp_f64<0x140, 0xF>(v));
  v = fmax(v, dpp_f64<0x142, 0xA>(v));
  v = fmax(v, dpp_f64<0x143, 0xC>(v));
  return readlane63(v);
}
// Sums eight per-lane doubles over the wave at once.  v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves
// and 16-lane rows between two registers, so one swap + one add halves the lanes of TWO quantities; the last four
// steps are the in-row DPP butterfly.  ~45 vector instructions for eight sums instead of ~20 each with wave_sum.
// On return the 16 lanes of row r hold total[map(r)] in lo (values 0..3) and hi (values 4..7), map = {0, 2, 1, 3}.
// Fixed combination order: deterministic.
__device__ __forceinline__ void swap32(double& a, double& b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap16(double& a, double& b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double row_sum16(double v) {  // all 16 lanes of a row end with the row's sum
  v += dpp_f64<0xB1, 0xF>(v);
  v += dpp_f64<0x4E, 0xF>(v);
  v += dpp_f64<0x141, 0xF>(v);
  v += dpp_f64<0x140, 0xF>(v);
  return v;
}
__device__ __forceinline__ void wave_sum8(double (&v)[8], double& lo, double& hi) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    swap32(v[i], v[i + 1]);
    v[i] += v[i + 1];          // lanes 0-31: value i over both halves; lanes 32-63: value i+1
  }
  swap16(v[0], v[2]);
  swap16(v[4], v[6]);
  lo = row_sum16(v[0] + v[2]);  // rows: values 0, 2, 1, 3
  hi = row_sum16(v[4] + v[6]);  // rows: values 4, 6, 5, 7
}

// The in-row half of those reductions on the LDS crossbar instead of the VALU: ds_swizzle (bit mode: lane ^ x inside a
// group of 32 lanes) moves the partner's value without touching memory, so a butterfly step costs the vector unit one
// add instead of two DPP moves and an add.  The row pass is bound by vector issue and its LDS pipe has room.
template <int XOR>
__device__ __forceinline__ double swz_xor_f64(double v) {
  constexpr int pat = 0x1f | (XOR << 10);   // and_mask 0x1f, or_mask 0, xor_mask XOR
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16_swz(double v) {  // all 16 lanes of a row end with the row's sum
  v += swz_xor_f64<1>(v);
  v += swz_xor_f64<2>(v);
  v += swz_xor_f64<4>(v);
  v += swz_xor_f64<8>(v);
  return v;
}
// same combination order as wave_sum8 (swap32, swap16, then xor 1, 2, 4, 8 inside a row): bit-identical totals
__device__ __forceinline__ void wave_sum8_swz(double (&v)[8], double& lo, double& hi) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    swap32(v[i], v[i + 1]);
    v[i] += v[i + 1];
  }
  swap16(v[0], v[2]);
  swap16(v[4], v[6]);
  lo = row_sum16_swz(v[0] + v[2]);
  hi = row_sum16_swz(v[4] + v[6]);
}
// maximum of non-negative values over the wave; lanes 32..63 end with it.  v_max_f64 itself (no canonicalising
// self-maximum in front: the inputs are sums of |z| maxima, never signalling).
__device__ __forceinline__ double vmax_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));   // the nops: a DPP read may follow
  return r;
}
__device__ __forceinline__ double wave_max_swz_hi(double v) {
  v = vmax_raw(v, swz_xor_f64<1>(v));
  v = vmax_raw(v, swz_xor_f64<2>(v));
  v = vmax_raw(v, swz_xor_f64<4>(v));
  v = vmax_raw(v, swz_xor_f64<8>(v));
  v = vmax_raw(v, swz_xor_f64<16>(v));
  // lanes 0..31 hold the maximum of the lower half, lanes 32..63 of the upper: lane 31 -> rows 2, 3 (0 elsewhere)
  return vmax_raw(v, dpp_f64<0x143, 0xC>(v));
}

__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// opcode semantics (codes/funcs.py:179-212)
template <typename T> __device__ __forceinline__ T op_exp(T x);
// exp is evaluated unconditionally and the clip applied by a select (NaN <= 200 is false: 1e10, as in the reference's
// per-element loop): a lane-divergent branch around it would turn the interpreter loop into a structurized region
// (see sin_rows below).  fp64: the table-driven bsr_exp (bsr_fastmath.h), which clamps its argument itself.
template <> __device__ __forceinline__ double op_exp<double>(double x) {
  const double e = bsr_exp(x, bsr_lds_tab);
  return (x <= 200.0) ? e : 1e10;
}
template <> __device__ __forceinline__ float op_exp<float>(float x) {
  const float e = expf(fminf(x, 200.0f));
  return (x <= 200.0f) ? e : 1e10f;
}
// sin/cos carry lane-divergent branches (large-argument reduction).  Inlined into the interpreter they make the whole
// node loop a divergent region, which LLVM then structurizes into long chains of flag-guarded blocks (~150 scalar
// instructions per node).  Kept out of line, the loop has only wave-uniform branches and stays a plain scalar
// switch; the call costs a few dozen cycles against the ~85 fp64 instructions of the function itself.
template <typename T, int U>
struct VecOf;
template <>
struct VecOf<double, 2> { using type = double2; };
template <>
struct VecOf<float, 2> { using type = float2; };
template <>
struct VecOf<double, 4> { using type = double4; };
template <>
struct VecOf<float, 4> { using type = float4; };
// fp64: branch-free bsr_sincos (bsr_fastmath.h).  When some lane of the wave holds a huge, infinite or NaN argument --
// a wave-uniform test -- the wave also runs the library routine and THOSE lanes take its result: every value stays
// a function of its own argument alone, whichever rows share its wave (the derived columns of bsr_api.hip and an
// inline evaluation of the same node must agree to the bit).
static __device__ __forceinline__ double sincos_big(double x, double fast, int which) {
  const double lib = which ? cos(x) : sin(x);
  return (fabs(x) < BSR_SINCOS_LIMIT) ? fast : lib;
}
// The same without a call on the ordinary path.  A kernel with LDS-DMA copies in flight cannot afford calls in its hot
// loop: every device function starts with s_waitcnt vmcnt(0) (the calling convention), which waits for the copies the
// wave issued for chunks it will not touch for microseconds.  The huge / non-finite lanes still go out of line: then, and
// only then, the wave pays that wait.
static __device__ __attribute__((noinline)) double sincos_big_call(double x, double fast, int which) {
  return sincos_big(x, fast, which);
}
template <int U>
__device__ __forceinline__ void sincos_vals(double (&v)[U], int which) {
  bool small = true;
  double f[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    small = small && (fabs(v[u]) < BSR_SINCOS_LIMIT);   // false for NaN
    f[u] = bsr_sincos(v[u], which, bsr_lds_tab);
  }
  if (__builtin_amdgcn_ballot_w64(!small) != 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = sincos_big_call(v[u], f[u], which);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = f[u];
}
static __device__ __attribute__((noinline)) double2 sin_rows(double2 v) {
  const bool small = fabs(v.x) < BSR_SINCOS_LIMIT && fabs(v.y) < BSR_SINCOS_LIMIT;  // false for NaN
  const double2 f = make_double2(bsr_sincos(v.x, 0, bsr_lds_tab), bsr_sincos(v.y, 0, bsr_lds_tab));
  if (__builtin_amdgcn_ballot_w64(!small) == 0) return f;
  return make_double2(sincos_big(v.x, f.x, 0), sincos_big(v.y, f.y, 0));
}
static __device__ __attribute__((noinline)) double2 cos_rows(double2 v) {
  const bool small = fabs(v.x) < BSR_SINCOS_LIMIT && fabs(v.y) < BSR_SINCOS_LIMIT;
  const double2 f = make_double2(bsr_sincos(v.x, 1, bsr_lds_tab), bsr_sincos(v.y, 1, bsr_lds_tab));
  if (__builtin_amdgcn_ballot_w64(!small) == 0) return f;
  return make_double2(sincos_big(v.x, f.x, 1), sincos_big(v.y, f.y, 1));
}
static __device__ __attribute__((noinline)) double2 exp_rows(double2 v) {
  return make_double2(op_exp<double>(v.x), op_exp<double>(v.y));
}
static __device__ __attribute__((noinline)) float2 exp_rows(float2 v) {
  return make_float2(op_exp<float>(v.x), op_exp<float>(v.y));
}
static __device__ __attribute__((noinline)) float2 sin_rows(float2 v) { return make_float2(sinf(v.x), sinf(v.y)); }
static __device__ __attribute__((noinline)) float2 cos_rows(float2 v) { return make_float2(cosf(v.x), cosf(v.y)); }
// The table extensions (log, div) are out of line for a different reason: inlined, their temporaries push the
// interpreter of the tile pass (128 VGPRs at 16 waves per CU) into scratch spills on every tape, used or not.
static __device__ __attribute__((noinline)) double2 log_rows(double2 v) {
  return make_double2(v.x == 0.0 ? 0.0 : log(fabs(v.x)), v.y == 0.0 ? 0.0 : log(fabs(v.y)));
}
static __device__ __attribute__((noinline)) float2 log_rows(float2 v) {
  return make_float2(v.x == 0.0f ? 0.0f : (float)log(fabs((double)v.x)), v.y == 0.0f ? 0.0f : (float)log(fabs((double)v.y)));
}
static __device__ __attribute__((noinline)) double2 div_rows(double2 l, double2 r) {  // protected like inv
  return make_double2(r.x == 0.0 ? 0.0 : l.x / r.x, r.y == 0.0 ? 0.0 : l.y / r.y);
}
static __device__ __attribute__((noinline)) float2 div_rows(float2 l, float2 r) {
  return make_float2(r.x == 0.0f ? 0.0f : l.x / r.x, r.y == 0.0f ? 0.0f : l.y / r.y);
}

// Four values per call for the interpreter's two-block passes: arguments and results then occupy the same registers of
// the call ABI, and nothing of the node being evaluated has to survive a call.
static __device__ __attribute__((noinline)) double4 sin_rows(double4 v) {
  const bool small = fabs(v.x) < BSR_SINCOS_LIMIT && fabs(v.y) < BSR_SINCOS_LIMIT && fabs(v.z) < BSR_SINCOS_LIMIT &&
                     fabs(v.w) < BSR_SINCOS_LIMIT;
  const double4 f = make_double4(bsr_sincos(v.x, 0, bsr_lds_tab), bsr_sincos(v.y, 0, bsr_lds_tab),
                                 bsr_sincos(v.z, 0, bsr_lds_tab), bsr_sincos(v.w, 0, bsr_lds_tab));
  if (__builtin_amdgcn_ballot_w64(!small) == 0) return f;
  return make_double4(sincos_big(v.x, f.x, 0), sincos_big(v.y, f.y, 0), sincos_big(v.z, f.z, 0), sincos_big(v.w, f.w, 0));
}
static __device__ __attribute__((noinline)) double4 cos_rows(double4 v) {
  const bool small = fabs(v.x) < BSR_SINCOS_LIMIT && fabs(v.y) < BSR_SINCOS_LIMIT && fabs(v.z) < BSR_SINCOS_LIMIT &&
                     fabs(v.w) < BSR_SINCOS_LIMIT;
  const double4 f = make_double4(bsr_sincos(v.x, 1, bsr_lds_tab), bsr_sincos(v.y, 1, bsr_lds_tab),
                                 bsr_sincos(v.z, 1, bsr_lds_tab), bsr_sincos(v.w, 1, bsr_lds_tab));
  if (__builtin_amdgcn_ballot_w64(!small) == 0) return f;
  return make_double4(sincos_big(v.x, f.x, 1), sincos_big(v.y, f.y, 1), sincos_big(v.z, f.z, 1), sincos_big(v.w, f.w, 1));
}
static __device__ __attribute__((noinline)) double4 exp_rows(double4 v) {
  return make_double4(op_exp<double>(v.x), op_exp<double>(v.y), op_exp<double>(v.z), op_exp<double>(v.w));
}
static __device__ __attribute__((noinline)) double4 log_rows(double4 v) {
  return make_double4(v.x == 0.0 ? 0.0 : log(fabs(v.x)), v.y == 0.0 ? 0.0 : log(fabs(v.y)),
                      v.z == 0.0 ? 0.0 : log(fabs(v.z)), v.w == 0.0 ? 0.0 : log(fabs(v.w)));
}
static __device__ __attribute__((noinline)) double4 div_rows(double4 l, double4 r) {
  return make_double4(r.x == 0.0 ? 0.0 : l.x / r.x, r.y == 0.0 ? 0.0 : l.y / r.y, r.z == 0.0 ? 0.0 : l.z / r.z,
                      r.w == 0.0 ? 0.0 : l.w / r.w);
}
static __device__ __attribute__((noinline)) float4 sin_rows(float4 v) { return make_float4(sinf(v.x), sinf(v.y), sinf(v.z), sinf(v.w)); }
static __device__ __attribute__((noinline)) float4 cos_rows(float4 v) { return make_float4(cosf(v.x), cosf(v.y), cosf(v.z), cosf(v.w)); }
static __device__ __attribute__((noinline)) float4 exp_rows(float4 v) {
  return make_float4(op_exp<float>(v.x), op_exp<float>(v.y), op_exp<float>(v.z), op_exp<float>(v.w));
}
static __device__ __attribute__((noinline)) float4 log_rows(float4 v) {
  return make_float4(v.x == 0.0f ? 0.0f : (float)log(fabs((double)v.x)), v.y == 0.0f ? 0.0f : (float)log(fabs((double)v.y)),
                     v.z == 0.0f ? 0.0f : (float)log(fabs((double)v.z)), v.w == 0.0f ? 0.0f : (float)log(fabs((double)v.w)));
}
static __device__ __attribute__((noinline)) float4 div_rows(float4 l, float4 r) {
  return make_float4(r.x == 0.0f ? 0.0f : l.x / r.x, r.y == 0.0f ? 0.0f : l.y / r.y, r.z == 0.0f ? 0.0f : l.z / r.z,
                     r.w == 0.0f ? 0.0f : l.w / r.w);
}

// acc <- f(acc) through the out-of-line routine `f`, four values at a time where the pass holds a multiple of four
#define BSR_CALL_ROWS(f)                                                           \
  if constexpr (U % 4 == 0 && QUAD) {                                                      \
    _Pragma("unroll") for (int j = 0; j < U / 4; ++j) {                            \
      typename VecOf<T, 4>::type r, v;                                             \
      v.x = acc[4 * j]; v.y = acc[4 * j + 1]; v.z = acc[4 * j + 2]; v.w = acc[4 * j + 3]; \
      r = f(v);                                                                    \
      acc[4 * j] = r.x; acc[4 * j + 1] = r.y; acc[4 * j + 2] = r.z; acc[4 * j + 3] = r.w; \
    }                                                                              \
  } else {                                                                         \
    _Pragma("unroll") for (int j = 0; j < U / 2; ++j) {                            \
      typename VecOf<T, 2>::type r, v;                                             \
      v.x = acc[2 * j]; v.y = acc[2 * j + 1];                                      \
      r = f(v);                                                                    \
      acc[2 * j] = r.x; acc[2 * j + 1] = r.y;                                      \
    }                                                                              \
  }

// np.power(x, 3) is libm pow (error < 1 ulp, in practice correctly rounded); x*x*x carries two roundings.
// Compensated product: x^3 = (x2 + e) * x with x2 + e == x*x exactly, rounded once at the end.
template <typename T> __device__ __forceinline__ T op_cube(T x);
template <> __device__ __forceinline__ double op_cube<double>(double x) {
  const double x2 = x * x;
  const double e = fma(x, x, -x2);
  const double p = x2 * x;
  const double pe = fma(x2, x, -p);
  const double r = p + (pe + e * x);
  // keep inf/NaN and overflow behaviour of the plain product: a non-finite p makes pe (and so r) NaN, and an r that
  // overflowed next to the largest double falls back to the finite p -- one test covers both
  return isfinite(r) ? r : p;
}
template <> __device__ __forceinline__ float op_cube<float>(float x) {
  const double xd = (double)x;
  return (float)(xd * xd * xd);
}

// Register stack accessed with a wave-uniform switch: no dynamic VGPR indexing, no scratch.  S slots live in VGPRs,
// deeper entries (Strahler number of the tree > S+1, rare) go to a per-wave global spill area.
// SPILL = false: the caller guarantees that no tape needs more than S slots (the deeper path -- vector memory -- is not
// even generated: a kernel that counts its own LDS-DMA copies must not find loads of the compiler's among them).
template <typename T, int U, int S, bool SPILL = true>
struct RegStack {
  T s[S][U];
  T* spill;  // per-wave: slot-major [slot][64*U]
  int lane;
  __device__ __forceinline__ void push(int sp, const T (&v)[U]) {
    if (sp < S) {
      switch (sp) {
#define X(i)                                                             \
  case i:                                                                \
    if constexpr (i < S) {                                               \
      _Pragma("unroll") for (int u = 0; u < U; ++u) s[i][u] = v[u];      \
    } else {                                                             \
      __builtin_unreachable();                                           \
    }                                                                    \
    break;
        X(0) X(1) X(2) X(3) X(4) X(5) X(6)
#undef X
        default: __builtin_unreachable();
      }
    } else if constexpr (SPILL) {
      T* q = spill + (size_t)(sp - S) * (BSR_WAVE * U) + lane * U;
#pragma unroll
      for (int u = 0; u < U; ++u) q[u] = v[u];
    } else {
      __builtin_unreachable();
    }
  }
  __device__ __forceinline__ void pop(int sp, T (&v)[U]) {
    if (sp < S) {
      switch (sp) {
#define X(i)                                                             \
  case i:                                                                \
    if constexpr (i < S) {                                               \
      _Pragma("unroll") for (int u = 0; u < U; ++u) v[u] = s[i][u];      \
    } else {                                                             \
      __builtin_unreachable();  /* otherwise: v undefined -> a zero the compiler materialises every pass */ \
    }                                                                    \
    break;
        X(0) X(1) X(2) X(3) X(4) X(5) X(6)
#undef X
        default: __builtin_unreachable();
      }
    } else if constexpr (SPILL) {
      const T* q = spill + (size_t)(sp - S) * (BSR_WAVE * U) + lane * U;
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = q[u];
    } else {
      __builtin_unreachable();
    }
  }
};

// Terminal loaders.  A lane owns U/2 pairs of adjacent rows, pair j at chunk offset j*128 + 2*lane, so every
// 16-byte access of the wave covers 1 KiB of consecutive addresses (LDS: conflict-free ds_read_b128).
// LdsCols reads the workgroup's staged tile [slot][rb_rows]; GlobalCols reads the feature-major matrix directly.
template <typename T, int U>
struct LdsCols {
  const T* sx;
  int rb_rows;
  int off;  // sweep offset + 2*lane, inside the tile
  __device__ __forceinline__ void load(int slot, T (&v)[U]) const {
    const T* col = sx + slot * rb_rows + off;
#pragma unroll
    for (int j = 0; j < U / 2; ++j) {
      v[2 * j] = col[j * 128];
      v[2 * j + 1] = col[j * 128 + 1];
    }
  }
};
template <typename T, int U>
struct GlobalCols {
  const T* __restrict__ Xt;
  int64_t ld, r0;  // r0 = absolute row of the lane's first pair
  __device__ __forceinline__ void load(int feature, T (&v)[U]) const {
    const T* col = Xt + (int64_t)feature * ld + r0;
#pragma unroll
    for (int j = 0; j < U / 2; ++j) {
      v[2 * j] = col[j * 128];
      v[2 * j + 1] = col[j * 128 + 1];
    }
  }
};

// XCD-aware work mapping for the row passes.  Workgroups are dealt round-robin over the 8 XCDs (id % 8 labels the
// XCD's group), each with a private 4 MiB L2.  All proposal groups of one row block, and a fixed eighth of the row
// blocks, go to the same group, so an XCD's L2 only ever holds its 1/8 slice of X, y and the cached basis columns
// (1.7 MB at N=100k, d=10, K=3) instead of all of it.  Placement only affects speed, never results.
struct WorkItem {
  int rb, pgi;
  bool valid;
};
__device__ __forceinline__ WorkItem map_work(int n_rb, int n_pg, int w) {
  const int xcd = w & 7, j = w >> 3;
  WorkItem it;
  it.pgi = j % n_pg;
  it.rb = (j / n_pg) * 8 + xcd;
  it.valid = it.rb < n_rb;
  return it;
}

// Evaluates one tape on the lane's U rows.  The tape arrives as three compact streams built by the host from the
// bsr_node rows (bsr_stage.hip: stage_tapes): 4-bit opcodes (16 per 64-bit word; `terminal, +|*` pairs arrive fused as
// BSR_SOP_ADD_T / BSR_SOP_MUL_T), 16-bit column ids of the terminals
// in tape order (4 per word) and the (a,b) pairs of the ln nodes.  They are read through the constant address space
// (scalar loads) a whole word at a time, so the node loop itself is register-only: opcode, stack pointer and every
// branch are wave-uniform, and the only memory operation on the critical path is the terminal read, which is
// requested one terminal ahead.  Each stream is padded so that reading one element past the end is always valid.
// The first words of a tape's three streams (wave-uniform: they live in SGPRs).  A caller that runs one tape over many
// row blocks loads them once (load_tape_head) instead of paying three scalar round trips per block.
struct TapeHead {
  uint64_t code0, code1, f0, f1;
  double la, lb;            // the (a, b) pair of the tape's first ln node
  const double* ln_near;    // three pairs (the first one again) in the tape's record, or null: the ln stream only
  int n_ln, n_term;         // ln nodes / terminals of the tape, or -1: not known (the streams are then read ahead
                            // unconditionally, as far as their padding allows)
};
__device__ __forceinline__ TapeHead load_tape_head(const uint64_t* codes, const uint64_t* feats, const double* lnp) {
  const uint64_t CONSTANT_AS* cw = as_const(codes);
  const uint64_t CONSTANT_AS* fw = as_const(feats);
  const double CONSTANT_AS* lp = as_const(lnp);
  TapeHead h;
  h.code0 = cw[0];
  h.code1 = cw[1];
  h.f0 = fw[0];
  h.f1 = fw[1];
  h.la = lp[0];
  h.lb = lp[1];
  h.ln_near = nullptr;
  h.n_ln = -1;
  h.n_term = -1;
  return h;
}
// The (a, b) pairs of a tape's ln nodes in stream order, fetched one ahead of their use.  The first three travel with
// the tape's record (tile pass: the 128 bytes the wave has just read -- a hit in the scalar cache); later ones, and all
// of them where there is no record, come from the ln stream: memory the host wrote microseconds ago, a miss in every
// cache, and a wait that also holds up the LDS reads queued behind it.  A tape of the real mix (at most three ln
// nodes, eight terminals, 32 entries) never goes there.
struct LnFeed {
  double a, b;
  int used, n_ln;
  const double CONSTANT_AS* lp;
  const double CONSTANT_AS* near;
  __device__ __forceinline__ void init(const TapeHead& hd, const double* lnp) {
    a = hd.la;
    b = hd.lb;
    used = 0;
    n_ln = hd.n_ln;
    lp = as_const(lnp);
    near = as_const(hd.ln_near);
  }
  __device__ __forceinline__ void next() {   // the pair in (a, b) has been used
    ++used;
    if (hd_near() && used < 3) {
      a = near[2 * used];
      b = near[2 * used + 1];
    } else if (n_ln < 0 || used < n_ln) {
      a = lp[2 * used];
      b = lp[2 * used + 1];
    }
  }
  __device__ __forceinline__ bool hd_near() const { return near != nullptr; }
};

// QUAD: the out-of-line routines take four values per call (a caller short of registers asks for pairs).
// INL (fp64): sin, cos, exp and the protected division inline -- for callers that must not call (sincos_vals above).
template <typename T, int U, int S, typename Loader, bool QUAD = true, bool INL = false>
__device__ __forceinline__ void run_tape_head(const TapeHead& hd, const uint64_t* codes, const uint64_t* feats,
                                              const double* lnp, int n, const Loader& ldr, T (&acc)[U], T* spill,
                                              int lane) {
  const uint64_t CONSTANT_AS* cw = as_const(codes);
  const uint64_t CONSTANT_AS* fw = as_const(feats);
  RegStack<T, U, S, !INL> st;   // (INL callers stay within the register stack)
  st.spill = spill;
  st.lane = lane;
  uint64_t code = hd.code0, code_next = hd.code1;
  uint64_t fhead = hd.f0, fnext = hd.f1;
  LnFeed ln;
  ln.init(hd, lnp);
  int ci = 1, fi = 1, nt = 0;
  int sp = 0;  // values on the stack below the accumulator
  T pre[U];
  ldr.load((int)(fhead & 0xFFFFu), acc);  // node 0 is always a terminal
  fhead >>= 16;
  nt = 1;
  ldr.load((int)(fhead & 0xFFFFu), pre);  // data of the next terminal (padding repeats a valid column)
  code >>= 4;
  for (int i = 1; i < n; ++i) {
    if ((i & 15) == 0) {
      code = code_next;
      ++ci;
      if (16 * ci < n) code_next = cw[ci];
    }
    const int op = (int)(code & 15u);
    code >>= 4;
    if (op >= BSR_OP_SUB) {  // extensions beyond the reference's table (semantics: oracle/bsr_oracle.py allcal)
      if (op == BSR_OP_LOG) {
        BSR_CALL_ROWS(log_rows)
      } else {
        T lhs[U];
        --sp;
        st.pop(sp, lhs);
        if (op == BSR_OP_SUB) {
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = lhs[u] - acc[u];
        } else if constexpr (INL) {  // BSR_OP_DIV, protected like inv
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = (acc[u] == (T)0) ? (T)0 : lhs[u] / acc[u];
        } else {  // BSR_OP_DIV
          if constexpr (U % 4 == 0 && QUAD) {
#pragma unroll
            for (int j = 0; j < U / 4; ++j) {
              typename VecOf<T, 4>::type r, v, l;
              v.x = acc[4 * j]; v.y = acc[4 * j + 1]; v.z = acc[4 * j + 2]; v.w = acc[4 * j + 3];
              l.x = lhs[4 * j]; l.y = lhs[4 * j + 1]; l.z = lhs[4 * j + 2]; l.w = lhs[4 * j + 3];
              r = div_rows(l, v);
              acc[4 * j] = r.x; acc[4 * j + 1] = r.y; acc[4 * j + 2] = r.z; acc[4 * j + 3] = r.w;
            }
          } else {
#pragma unroll
            for (int j = 0; j < U / 2; ++j) {
              typename VecOf<T, 2>::type r, v, l;
              v.x = acc[2 * j];
              v.y = acc[2 * j + 1];
              l.x = lhs[2 * j];
              l.y = lhs[2 * j + 1];
              r = div_rows(l, v);
              acc[2 * j] = r.x;
              acc[2 * j + 1] = r.y;
            }
          }
        }
      }
    } else if (op >= BSR_OP_TERMINAL) {  // consumes the prefetched column
      if (op == BSR_OP_TERMINAL) {
        st.push(sp, acc);
        ++sp;
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = pre[u];
      } else if (op == BSR_SOP_ADD_T) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = acc[u] + pre[u];
      } else {  // BSR_SOP_MUL_T
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = acc[u] * pre[u];
      }
      fhead >>= 16;
      if (++nt == 4) {
        fhead = fnext;
        ++fi;
        if (hd.n_term < 0 || 4 * fi <= hd.n_term) fnext = fw[fi];   // (one terminal is requested ahead of its use)
        nt = 0;
      }
      ldr.load((int)(fhead & 0xFFFFu), pre);
    } else if (op >= BSR_OP_ADD) {
      T lhs[U];
      --sp;
      st.pop(sp, lhs);
      if (op == BSR_OP_ADD) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = lhs[u] + acc[u];
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = lhs[u] * acc[u];
      }
    } else {
      switch (op) {
        case BSR_OP_INV:
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = (acc[u] == (T)0) ? (T)0 : (T)1 / acc[u];
          break;
        case BSR_OP_LN: {
          const T a = (T)ln.a, b = (T)ln.b;
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = a * acc[u] + b;  // two roundings (contraction is off)
          ln.next();
        } break;
        case BSR_OP_NEG:
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = -acc[u];
          break;
        case BSR_OP_SIN:
          if constexpr (INL && sizeof(T) == 8) {
            sincos_vals<U>(acc, 0);
          } else {
            BSR_CALL_ROWS(sin_rows)
          }
          break;
        case BSR_OP_COS:
          if constexpr (INL && sizeof(T) == 8) {
            sincos_vals<U>(acc, 1);
          } else {
            BSR_CALL_ROWS(cos_rows)
          }
          break;
        case BSR_OP_EXP:
          if constexpr (INL && sizeof(T) == 8) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = op_exp<T>(acc[u]);
          } else {
            BSR_CALL_ROWS(exp_rows)
          }
          break;
        case BSR_OP_SQUARE:
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = acc[u] * acc[u];
          break;
        default:  // BSR_OP_CUBIC
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = op_cube<T>(acc[u]);
          break;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Chain tapes.  After the encoder's two fusions (`terminal f, unary op` -> derived column, `terminal, +|*` -> one entry;
// commutative operands reordered so that a bare terminal comes second) almost every tape of the real move mix is a
// CHAIN: its first entry is a terminal and every other entry maps the accumulator to the accumulator -- a unary
// operator or `acc (+|*) column`.  A chain needs no value stack, so a wave can hold a whole pass of NB row blocks
// (2 NB values per lane) in registers and run the tape over them entry by entry: the wave-uniform decode of an entry is
// paid once per NB * 128 rows, every operator works in place (no operand copies), and the values of a row are the
// same as the stack machine's (same operators, same order; a + b == b + a bit for bit).
// Values 2j, 2j+1 of a pass = the lane's pair of rows of block j.  FULL: the pass holds NB blocks (the hot case: every
// load has a compile-time offset); otherwise nb < NB of them, and the blocks behind copy block 0 -- computed along,
// never accumulated.
template <typename T, int NB, bool FULL>
struct LdsPass {
  const T* sx;
  int rb_rows;
  int off;  // element offset of the lane's pair in block 0 of the pass, inside the staged slice
  int nb;   // blocks of this pass, 1..NB (wave-uniform)
  __device__ __forceinline__ void load(int slot, T (&v)[2 * NB]) const {
    using V2 = typename VecOf<T, 2>::type;
    const T* col = sx + slot * rb_rows + off;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (FULL || j == 0 || j < nb) {
        const V2 p = *reinterpret_cast<const V2*>(col + j * 128);
        v[2 * j] = p.x;
        v[2 * j + 1] = p.y;
      } else {
        v[2 * j] = v[0];
        v[2 * j + 1] = v[1];
      }
    }
  }
};

// acc <- f(acc) through the out-of-line routine `f`, four values per call; groups that hold no block of the pass are skipped
#define BSR_CHAIN_CALL(f)                                                          \
  _Pragma("unroll") for (int j = 0; j < U / 4; ++j) {                              \
    if (FULL || j == 0 || 2 * j < nb) {                                            \
      typename VecOf<T, 4>::type r, v;                                             \
      v.x = acc[4 * j]; v.y = acc[4 * j + 1]; v.z = acc[4 * j + 2]; v.w = acc[4 * j + 3]; \
      r = f(v);                                                                    \
      acc[4 * j] = r.x; acc[4 * j + 1] = r.y; acc[4 * j + 2] = r.z; acc[4 * j + 3] = r.w; \
    }                                                                              \
  }

template <typename T, int NB, bool FULL>
__device__ __forceinline__ void chain_eval(const TapeHead& hd, const uint64_t* codes, const uint64_t* feats,
                                           const double* lnp, int n, const T* sx, int rb_rows, int off, int nb,
                                           T (&acc)[2 * NB]) {
  constexpr int U = 2 * NB;
  static_assert(U % 4 == 0, "the out-of-line routines take four values");
  const uint64_t CONSTANT_AS* cw = as_const(codes);
  const uint64_t CONSTANT_AS* fw = as_const(feats);
  const LdsPass<T, NB, FULL> ldr{sx, rb_rows, off, nb};
  uint64_t code = hd.code0, code_next = hd.code1;
  uint64_t fhead = hd.f0, fnext = hd.f1;
  LnFeed ln;
  ln.init(hd, lnp);
  int ci = 1, fi = 1, nt = 1;
  ldr.load((int)(fhead & 0xFFFFu), acc);  // entry 0 is the chain's terminal
  fhead >>= 16;
  code >>= 4;
  for (int i = 1; i < n; ++i) {
    if ((i & 15) == 0) {
      code = code_next;
      ++ci;
      if (16 * ci < n) code_next = cw[ci];
    }
    const int op = (int)(code & 15u);
    code >>= 4;
    switch (op) {
      case BSR_SOP_ADD_T:
      case BSR_SOP_MUL_T: {
        T pre[U];
        ldr.load((int)(fhead & 0xFFFFu), pre);
        fhead >>= 16;
        if (++nt == 4) {
          fhead = fnext;
          ++fi;
          if (hd.n_term < 0 || 4 * fi < hd.n_term) fnext = fw[fi];
          nt = 0;
        }
        if (op == BSR_SOP_ADD_T) {
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = acc[u] + pre[u];
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = acc[u] * pre[u];
        }
      } break;
      case BSR_OP_INV:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = (acc[u] == (T)0) ? (T)0 : (T)1 / acc[u];
        break;
      case BSR_OP_LN: {
        const T a = (T)ln.a, b = (T)ln.b;
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = a * acc[u] + b;  // two roundings (contraction is off)
        ln.next();
      } break;
      case BSR_OP_NEG:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = -acc[u];
        break;
      case BSR_OP_SIN:   // out of line: inlined, the constants and temporaries of sixteen values spill (measured: 16 VGPRs,
        BSR_CHAIN_CALL(sin_rows)   // 115 SGPRs), and a call costs ~3 register moves per value next to ~31 instructions
        break;
      case BSR_OP_COS:
        BSR_CHAIN_CALL(cos_rows)
        break;
      case BSR_OP_EXP:
        BSR_CHAIN_CALL(exp_rows)
        break;
      case BSR_OP_LOG:
        BSR_CHAIN_CALL(log_rows)
        break;
      case BSR_OP_SQUARE:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = acc[u] * acc[u];
        break;
      case BSR_OP_CUBIC:
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = op_cube<T>(acc[u]);
        break;
      default:  // a chain holds no other entry (the host routes everything else to the stack machine)
        break;
    }
  }
}

template <typename T, int U, int S, typename Loader>
__device__ __forceinline__ void run_tape(const uint64_t* codes, const uint64_t* feats, const double* lnp, int n,
                                         const Loader& ldr, T (&acc)[U], T* spill, int lane) {
  const TapeHead hd = load_tape_head(codes, feats, lnp);
  run_tape_head<T, U, S>(hd, codes, feats, lnp, n, ldr, acc, spill, lane);
}
