"""bsr: MI355X-native drop-in for the likelihood hot path of ying531/MCMC-SymReg.

Same package name and re-exports as the reference (setup.py:14-20, codes/__init__.py:9-12).  The O(N) work --
tree evaluation, OLS fit across the K trees, Gaussian log-likelihood, rank gate -- runs in hand-written HIP kernels
behind the C ABI of include/bsr_hip.h; this package is the Python host side.
"""
from .funcs import Operator, Node
from .funcs import grow, genList, shrink, upgOd, allcal, display, getHeight, getNum, numLT, upDepth, Express, fStruc
from .funcs import ylogLike, newProp, Prop, auxProp
from .bsr_class import BSR
