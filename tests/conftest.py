"""pytest configuration: markers, import paths and golden-fixture helpers."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "mcmc-symreg_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def unf(v):
    """Decode the fixture float encoding (None stays None; 'nan'/'inf'/'-inf' strings)."""
    if isinstance(v, str):
        return float(v)
    return v


def farr(lst):
    return np.array([np.nan if v is None else unf(v) for v in lst], dtype=np.float64)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rng_mark():
    import zlib
    st = np.random.get_state()
    return {"pos": int(st[2]), "crc": int(zlib.crc32(st[1].tobytes())), "has_gauss": int(st[3])}
