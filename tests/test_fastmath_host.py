"""Accuracy of the interpreter's own fp64 sin / cos / exp (mcmc-symreg_amd/csrc/bsr_fastmath.h) against 400-bit
references.  The header is plain C: gcc compiles it here exactly as hipcc does for the device (explicit fused
multiply-adds, contraction off), so the bounds pinned below are the device's."""
import os
import shutil
import subprocess

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mcmc-symreg_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    d = tmp_path_factory.mktemp("fastmath")
    exe = str(d / "fastmath_host")
    flags = ["-O2", "-ffp-contract=off", "-I", CSRC]
    try:
        if " fma " in open("/proc/cpuinfo").read():
            flags.append("-mfma")
    except OSError:
        pass
    subprocess.run(["gcc"] + flags + [os.path.join(ROOT, "tests", "helpers", "fastmath_host.c"), "-o", exe, "-lm"], check=True)

    def run(which, x):
        fi, fo = str(d / "in.bin"), str(d / "out.bin")
        np.asarray(x, dtype=np.float64).tofile(fi)
        subprocess.run([exe, str(which), fi, fo], check=True)
        return np.fromfile(fo, dtype=np.float64)
    return run


def ulp_errors(fn, x, got):
    mp.mp.prec = 400
    errs = np.empty(len(x))
    for i, (xi, gi) in enumerate(zip(x, got)):
        want = fn(mp.mpf(float(xi)))
        w = float(want)
        if w == 0.0 or not np.isfinite(w):
            errs[i] = 0.0 if gi == w else np.inf
            continue
        u = np.spacing(abs(w))
        errs[i] = float(abs(mp.mpf(float(gi)) - want) / mp.mpf(float(u)))
    return errs


def sample_trig(rs, n):
    """Uniform on the interpreter's typical range, log-uniform magnitudes up to the fast path's limit, and the worst
    neighbourhoods: next to multiples of pi/2 (the result is the reduced argument itself) and of pi/256 (the
    reduction's rounding boundary)."""
    k = rs.randint(-2 ** 20, 2 ** 20, size=n // 4)
    near_zero = k * (np.pi / 2) * (1.0 + rs.uniform(-4, 4, size=k.size) * 2.0 ** -50)
    kb = rs.randint(-2 ** 20, 2 ** 20, size=n // 4)
    boundary = (kb + 0.5) * (np.pi / 128) * (1.0 + rs.uniform(-4, 4, size=kb.size) * 2.0 ** -52)
    logu = rs.choice([-1.0, 1.0], size=n // 4) * 10.0 ** rs.uniform(-12, np.log10(1647098.0), size=n // 4)
    return np.concatenate([rs.uniform(-40, 40, size=n // 4), logu, near_zero, boundary])


@pytest.mark.parametrize("which,fn", [(0, "sin"), (1, "cos")])
def test_sin_cos_within_one_ulp(harness, which, fn):
    rs = np.random.RandomState(11 + which)
    x = sample_trig(rs, 40_000)
    x = x[np.abs(x) < 1647099.0]
    got = harness(which, x)
    errs = ulp_errors(getattr(mp, fn), x, got)
    assert errs.max() < 1.0, (errs.max(), x[np.argmax(errs)])
    assert np.mean(errs) < 0.3
    # monotone bookkeeping the interpreter's parity tests rely on
    edge = np.array([0.0, -0.0, 1e-300, -1e-300, 2.0 ** -27, -2.0 ** -27])
    out = harness(which, edge)
    if which == 0:
        assert np.array_equal(out, edge) and np.signbit(out[1])
    else:
        assert np.all(out == 1.0)


def test_exp_within_one_ulp(harness):
    rs = np.random.RandomState(5)
    x = np.concatenate([rs.uniform(-30, 30, size=15_000), rs.uniform(-745, 200, size=15_000),
                        rs.choice([-1.0, 1.0], size=5_000) * 10.0 ** rs.uniform(-18, 0, size=5_000),
                        (rs.randint(-60000, 18000, size=5_000) + 0.5) * (np.log(2) / 64) * (1 + rs.uniform(-4, 4, size=5_000) * 2.0 ** -52)])
    got = harness(2, x)
    normal = x > -708.0                                    # below: subnormal results, rounded twice (pinned next)
    errs = ulp_errors(mp.exp, x[normal], got[normal])
    assert errs.max() < 1.0, (errs.max(), x[normal][np.argmax(errs)])
    assert np.mean(errs) < 0.3
    errs = ulp_errors(mp.exp, x[~normal], got[~normal])
    assert errs.max() <= 1.5
    edge = np.array([0.0, -0.0, 709.7, 710.0, 1e308, np.inf, -745.2, -760.0, -1e308, -np.inf])
    out = harness(2, edge)
    assert out[0] == 1.0 and out[1] == 1.0
    assert np.isfinite(out[2]) and np.all(np.isinf(out[3:6])) and np.all(out[6:] == 0.0)
