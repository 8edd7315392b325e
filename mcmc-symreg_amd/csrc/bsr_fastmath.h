// fp64 sin / cos / exp for the tree-eval interpreter (codes/funcs.py:186, 199-201: the clipped exp loop, np.sin, np.cos).
//
// The row pass is bound by vector-instruction issue, and 0.8 transcendental nodes per tape at ~50 (sin, cos) and ~35
// (exp) instructions per value were half of everything it issued.  These versions trade polynomial degree for a table
// lookup -- the LDS pipe is idle in that kernel -- and come to ~28 / ~24 instructions per value:
//
//   sin, cos:  k = rint(x * 128/pi);  r + rl = x - k * pi/128 as a double-double (pi/128 = P1 + P2 + P3; the first
//              step is an exact fused multiply-add, the second a double-double product and a compensated
//              subtraction, so the remainder keeps full relative accuracy next to multiples of pi/128), |r| <= pi/256;
//              with j = k mod 256 (cos: k + 64) and {S, C} = {sin, cos}(j*pi/128) from the table as double-doubles,
//                 sin(j*pi/128 + r) = S + (C*r + (S*(cos r - 1) + C*(sin r - r) + S_lo + C_lo*r + C*rl)),
//              cos r - 1 and sin r - r by three Taylor terms each (next terms < 2^-66).
//              Branch-free for |x| < 2^20 * pi/2; the caller sends anything larger, inf and NaN to the library
//              routine (a wave-uniform test).
//   exp:       k = rint(x * 64/ln2); r = x - k * ln2/64 (two fused multiply-adds, the first exact), |r| <= ln2/128;
//              exp(x) = 2^(k >> 6) * T[k & 63] * e^r,  T[j] = 2^(j/64) as a double-double, e^r - 1 by six Taylor terms.
//
// Measured against 400-bit references (tests/test_fastmath_host.py compiles this header on the host with gcc and
// hardware fma -- BSR_HD expands to `static inline` there, the table pointer is then the generated array itself):
// see the test for the bounds it pins.  Tables and split constants: bsr_tables.h (tools/gen_tables.py).
#pragma once

#ifndef BSR_HD
#ifdef __HIPCC__
#define BSR_HD __device__ __forceinline__
#else
#define BSR_HD static inline
#endif
#endif

// clamp helpers: on the device a plain v_max_f64 / v_min_f64 (fmax/fmin would each add a canonicalising self-maximum)
#ifdef __HIPCC__
__device__ __forceinline__ double bsr_vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double bsr_vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#else
static inline double bsr_vmax(double a, double b) { return __builtin_fmax(a, b); }
static inline double bsr_vmin(double a, double b) { return __builtin_fmin(a, b); }
#endif

// double -> int32 of an integral value.  On the device the instruction itself (saturating; NaN -> 0): the fast paths
// also run on lanes whose huge or non-finite argument the caller replaces afterwards, where a C++ cast is undefined.
#ifdef __HIPCC__
__device__ __forceinline__ int bsr_cvt_i32(double k) {
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(k));
  return r;
}
#else
static inline int bsr_cvt_i32(double k) { return (k >= -2147483648.0 && k <= 2147483647.0) ? (int)k : 0; }
#endif

#define BSR_SINCOS_LIMIT 1647099.0   /* 2^20 * pi/2: beyond it the caller uses the library routine */

// which = 0: sin(x), which = 1: cos(x); tab = the 1152-double table block (bsr_tables.h layout)
BSR_HD double bsr_sincos(double x, int which, const double* tab) {
  const double k = __builtin_rint(x * BSR_TRIG_INV_STEP);
  const double rh = __builtin_fma(-k, BSR_TRIG_P1, x);      /* exact: 52 bits between 2^-6 and the last bit of k*P1 */
  const double ph = k * BSR_TRIG_P2;
  const double pl = __builtin_fma(k, BSR_TRIG_P2, -ph);      /* ph + pl == k*P2 exactly */
  const double r = rh - ph;
  const double e = (rh - r) - ph;                            /* what the rounding of r dropped (exact also when rh and
                                                                ph cancel: x next to a multiple of pi/128) */
  const double rl = __builtin_fma(-k, BSR_TRIG_P3, e - pl);
  const int j = (bsr_cvt_i32(k) + (which << 6)) & 255;
  const double* t = tab + 4 * j;
  const double S = t[0], Sl = t[1], C = t[2], Cl = t[3];
  const double z = r * r;
  /* sin r - r = r * ps,  cos r - 1 = pc */
  const double ps = z * __builtin_fma(z, __builtin_fma(z, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0);
  const double pc = z * __builtin_fma(z, __builtin_fma(z, -1.0 / 720.0, 1.0 / 24.0), -0.5);
  double u = S * pc;
  u = __builtin_fma(C, r * ps, u);
  u = u + Sl;
  u = __builtin_fma(Cl, r, u);
  u = __builtin_fma(C, rl, u);
  const double res = S + __builtin_fma(C, r, u);
  /* sin of a tiny argument is the argument itself (and keeps the sign of -0, which k = -0 would lose) */
  return (which == 0 && __builtin_fabs(x) < 0x1p-26) ? x : res;
}

// e^x for finite or infinite x.  A NaN comes out as e^710 = inf or e^-760 = 0 (v_min/v_max return the other operand):
// the interpreter's clipped exp (codes/funcs.py:186) maps NaN to its 1e10 branch before the value is looked at.
BSR_HD double bsr_exp(double x, const double* tab) {
  const double xc = bsr_vmax(bsr_vmin(x, 710.0), -760.0);   /* e^710 = inf, e^-760 = 0 after scaling */
  const double k = __builtin_rint(xc * BSR_EXP_INV_STEP);
  double r = __builtin_fma(-k, BSR_EXP_L1, xc);
  r = __builtin_fma(-k, BSR_EXP_L2, r);
  const int ki = bsr_cvt_i32(k);
  const double* t = tab + BSR_TRIG_TAB_DOUBLES + 2 * (ki & 63);
  const double T = t[0], Tl = t[1];
  double q = __builtin_fma(r, 1.0 / 720.0, 1.0 / 120.0);
  q = __builtin_fma(r, q, 1.0 / 24.0);
  q = __builtin_fma(r, q, 1.0 / 6.0);
  q = __builtin_fma(r, q, 0.5);
  const double p = __builtin_fma(r * r, q, r);            /* e^r - 1 */
  const double m = T + __builtin_fma(T, p, Tl);
  return __builtin_ldexp(m, ki >> 6);
}
