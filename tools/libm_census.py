#!/usr/bin/env python3
"""How far is the numpy of THIS container -- the one tests/golden/ was generated with -- from correct rounding, per
operation of codes/funcs.py:175-220?  (Build container only: needs mpmath.)

The reference evaluates exp per element through numpy's scalar path and sin/cos/power on whole arrays
(codes/funcs.py:182-205).  On an AVX-512 host numpy dispatches exp and power to its SIMD routines, which are not the C
library's and not correctly rounded; sin/cos go to glibc.  The device code (bsr_sincos.h, the device library's exp,
a compensated cube) is accurate to < 1 ulp but cannot be bit-equal to that build: the census below says how often
even a correctly rounded implementation would differ, i.e. the floor of any "bit-exact" claim for these opcodes."""
import math

import mpmath
import numpy as np

mpmath.mp.prec = 200


def census(name, np_fn, mp_fn, xs, libm=None, scalar=False):
    got = np.array([np_fn(v) for v in xs]) if scalar else np_fn(xs)
    bad = badm = 0
    for i, x in enumerate(xs):
        want = float(mp_fn(mpmath.mpf(float(x))))
        bad += got[i] != want
        if libm is not None:
            badm += libm(float(x)) != want
    print("%-34s n=%d  numpy != correctly rounded: %5d (%.2f %%)%s" %
          (name, len(xs), bad, 100.0 * bad / len(xs),
           "" if libm is None else "   C library != correctly rounded: %d (%.2f %%)" % (badm, 100.0 * badm / len(xs))))


def main():
    print("numpy", np.__version__, "| SIMD extensions found:",
          ",".join(np._core._multiarray_umath.__cpu_features__[k] and k or "" for k in ("AVX2", "AVX512F", "AVX512_SKX")).strip(","))
    rs = np.random.RandomState(1)
    n = 20000
    for lo, hi in ((-3, 3), (-30, 30), (-1e4, 1e4)):
        xs = rs.uniform(lo, hi, n)
        census("sin  x in [%g, %g]" % (lo, hi), np.sin, mpmath.sin, xs, math.sin)
        census("cos  x in [%g, %g]" % (lo, hi), np.cos, mpmath.cos, xs, math.cos)
    xs = rs.uniform(-20, 20, n)
    census("exp (scalar calls) x in [-20, 20]", np.exp, mpmath.exp, xs, math.exp, scalar=True)
    xs = rs.uniform(-3, 3, n)
    census("power(x, 3)  x in [-3, 3]", lambda v: np.power(v, 3), lambda v: v ** 3, xs, lambda v: math.pow(v, 3))
    census("square x in [-3, 3]", np.square, lambda v: v * v, xs)


if __name__ == "__main__":
    main()
