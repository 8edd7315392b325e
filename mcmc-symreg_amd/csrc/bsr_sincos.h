// fp64 sin / cos for the tree-eval interpreter (codes/funcs.py:199-201: np.sin / np.cos).
//
// Why not the device library's sin(): its large-argument path makes every call carry lane-divergent control flow and
// ~270 vector instructions per pair of values; in the tile row pass sin/cos were the largest single consumer of VALU
// issue slots.  This version is branch-free for |x| < 2^20 * pi/2 (a wave-uniform test sends anything larger, inf and
// NaN to the library routine): ~40 fp64 instructions per value.
//
//   k = rint(x * 2/pi);  r = x - k * pi/2 as a double-double (r, rl), pi/2 = P1 + P2 + P3 with fused multiply-adds:
//   the first fma is exact whenever 16 or more leading bits cancel and has relative error 2^-53 otherwise, the second
//   carries its rounding error into rl, so r + rl is good to ~2^-100 |x|;
//   then the two classic minimax kernels on [-pi/4, pi/4] with the tail correction (the published fdlibm / FreeBSD
//   msun __kernel_sin and __kernel_cos polynomials, error < 1 ulp); the quadrant k mod 4 selects and signs.
//
// The same source is compiled on the host by tests/ (gcc, with hardware fma) against long-double references:
// BSR_HD expands to nothing there.
#pragma once

#ifndef BSR_HD
#ifdef __HIPCC__
#define BSR_HD __device__ __forceinline__
#else
#define BSR_HD static inline
#endif
#endif

#define BSR_SINCOS_LIMIT 1647099.0   /* 2^20 * pi/2: beyond it the caller uses the library routine */

// which = 0: sin(x), which = 1: cos(x)
BSR_HD double bsr_sincos(double x, int which) {
  const double two_over_pi = 6.36619772367581382433e-01;   /* 0x3FE45F306DC9C883 */
  const double P1 = 1.57079632679489655800e+00;            /* 0x3FF921FB54442D18 */
  const double P2 = 6.12323399573676603587e-17;            /* 0x3C91A62633145C07 */
  const double P3 = -1.49738490485916983294e-33;           /* 0xB91F1976B7ED8FBC */
  const double k = __builtin_rint(x * two_over_pi);
  const double rh = __builtin_fma(-k, P1, x);
  const double r = __builtin_fma(-k, P2, rh);
  const double e = __builtin_fma(-k, P2, rh - r);          /* what the rounding of r dropped */
  const double rl = __builtin_fma(-k, P3, e);
  const int q = ((int)k + which) & 3;
  const double z = r * r;
  /* sin kernel */
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double v = z * r;
  const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
  const double sn = r - ((z * (0.5 * rl - v * ps) - rl) - v * S1);
  /* cos kernel */
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double pc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cs = w + (((1.0 - w) - hz) + (z * pc - r * rl));
  const double m = (q & 1) ? cs : sn;
  const double res = (q & 2) ? -m : m;
  /* sin of a tiny argument is the argument itself (and keeps the sign of -0, which k = -0 would lose) */
  return (which == 0 && __builtin_fabs(x) < 0x1p-26) ? x : res;
}
