"""Random draws and scalar densities of the sampler, bit-compatible with the reference's call sites.

The reference draws everything through numpy's global legacy RandomState (scipy's rvs included), so the
accepted-tree sequence for a seed is pinned by the ORDER and KIND of draws.  These shims consume
the global stream exactly like the reference's calls but skip the scipy.stats machinery:

  np.random.uniform(0,1,1)[0]            == random_sample()
  np.random.randint(lo,hi,1)[0]          == randint(lo,hi)            (same masked-rejection path)
  np.random.choice(arange(n), p=w)       == cdf.searchsorted(random_sample(), 'right')
  scipy.stats.norm.rvs(loc, scale)       == loc + scale*standard_normal()
  scipy.stats.invgamma.rvs(a)            == 1/gammainccinv(a, random_sample())
(equalities pinned by tests/test_oracle_golden.py::test_g7_rng_primitives and tests/test_host_driver.py)
"""
import math

import numpy as np
from scipy.special import gammainccinv

_rand = np.random.random_sample
_randint = np.random.randint
_normal = np.random.standard_normal

INF = float("inf")
NAN = float("nan")
LOG_SQRT_2PI = None
SQRT_2PI = math.sqrt(2 * math.pi)


def uniform():
    return _rand()


def randint(lo, hi):
    return int(_randint(lo, hi))


def randint_arr(lo, hi):
    """Shape-(1,) array like np.random.randint(lo, hi, 1): what the reference stores in Node.feature."""
    return _randint(lo, hi, 1)


class Chooser:
    """np.random.choice(np.arange(n), p=weights) without the per-call validation."""

    def __init__(self, weights):
        cdf = np.cumsum(np.asarray(weights, dtype=np.float64))
        cdf /= cdf[-1]
        self.cdf = cdf

    def __call__(self):
        return int(self.cdf.searchsorted(_rand(), side="right"))


def normal(loc, scale):
    return loc + scale * _normal()


def invgamma_rvs(a):
    return 1.0 / float(gammainccinv(a, _rand()))


# ---- IEEE-style scalar helpers (numpy semantics instead of Python exceptions)
def flog(x):
    if x > 0:
        return math.log(x)
    if x == 0:
        return -INF
    return NAN


def fexp(x):
    try:
        return math.exp(x)
    except OverflowError:
        return INF


def fdiv(a, b):
    if b != 0:
        return a / b
    if a != a or a == 0:
        return NAN
    return INF if (a > 0) == (math.copysign(1.0, b) > 0) else -INF


def invgamma_pdf(x, a):
    """scipy.stats.invgamma.pdf: exp(-(a+1) log x - lgamma(a) - 1/x)."""
    if not x > 0:
        return 0.0
    return fexp(-(a + 1) * math.log(x) - math.lgamma(a) - 1.0 / x)


def norm_pdf(x, loc, scale):
    """scipy.stats.norm.pdf(x, loc, scale) = exp(-z^2/2)/sqrt(2 pi)/scale."""
    z = (x - loc) / scale
    return math.exp(-z * z / 2.0) / SQRT_2PI / scale


def get_state():
    return np.random.get_state()


def set_state(st):
    np.random.set_state(st)
