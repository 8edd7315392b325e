"""CPU-only sanitizer jobs over the product's threaded host code (GPU sanitizers are not available on the pool).

The native sampler (csrc/bsr_engine.hip: worker threads, batches generated ahead, the context lock around the accept
path) is compiled as plain C++ against tests/native/stub_scorer.cpp -- a CPU stand-in for the data side of the C ABI,
test infrastructure only -- once with AddressSanitizer + UndefinedBehaviorSanitizer as a shared library that the
Python package loads instead of libbsr_hip.so, once with ThreadSanitizer around a small driver:
  * the reference's golden traces replay through the sanitized sampler bit for bit (the same test functions the GPU
    box runs against the HIP library);
  * eight chains on one, four and eight worker threads, with and without look-ahead batches, end in the same state,
    with no data race reported.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "native", "build_san.sh")


@pytest.fixture(scope="module")
def engine_host_exe(tmp_path_factory):
    """The native sampler against the CPU stand-in of the data side, plain -O2 build (one build for the tests that run it)."""
    exe = str(tmp_path_factory.mktemp("engine_host") / "engine_host")
    src = [os.path.join(ROOT, "mcmc-symreg_amd", "csrc", "bsr_engine.hip"), os.path.join(ROOT, "tests", "native", "stub_scorer.cpp"),
           os.path.join(ROOT, "tests", "native", "engine_tsan_main.cpp")]
    cmd = ["g++", "-std=c++17", "-O2", "-DBSR_HOST_ONLY", "-pthread", "-I" + os.path.join(ROOT, "include")]
    for f in src:
        cmd += ["-x", "c++", f]
    b = subprocess.run(cmd + ["-o", exe], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stderr[-4000:]
    return exe


def _runtime(name):
    out = subprocess.run(["gcc", "-print-file-name=%s" % name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_thread_sanitizer_finds_no_race_and_grouping_does_not_change_the_chains(tmp_path):
    if _runtime("libtsan.so") is None:
        pytest.skip("libtsan not installed")
    exe = str(tmp_path / "engine_tsan")
    b = subprocess.run(["bash", BUILD, "tsan", exe], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stderr[-4000:]
    outs = {}
    for tag, env in (("g1", {"BSR_ENGINE_GROUPS": "1"}), ("g4", {"BSR_ENGINE_GROUPS": "4"}),
                     ("g8", {"BSR_ENGINE_GROUPS": "8"}), ("g4_no_lookahead", {"BSR_ENGINE_GROUPS": "4", "BSR_ENGINE_LOOKAHEAD": "0"}),
                     ("g4_two_ahead", {"BSR_ENGINE_GROUPS": "4", "BSR_ENGINE_LOOKAHEAD": "2"}),
                     ("g1_one_ahead", {"BSR_ENGINE_GROUPS": "1", "BSR_ENGINE_LOOKAHEAD": "1"}),
                     # helper threads (a group's chains generated and consumed on threads of their own; the default deals
                     # one per group): none, and as many as the chains
                     ("g4_no_helpers", {"BSR_ENGINE_GROUPS": "4", "BSR_ENGINE_HELPERS": "0"}),
                     ("g2_six_helpers", {"BSR_ENGINE_GROUPS": "2", "BSR_ENGINE_HELPERS": "6", "BSR_ENGINE_LOOKAHEAD": "2"}),
                     ("g1_seven_helpers", {"BSR_ENGINE_GROUPS": "1", "BSR_ENGINE_HELPERS": "7"})):
        r = subprocess.run([exe, "8", "300"], env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0", **env),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (tag, r.stderr[-3000:])
        assert "ThreadSanitizer" not in r.stderr, (tag, r.stderr[:4000])
        outs[tag] = r.stdout
    assert len(outs["g1"].splitlines()) == 8
    assert len(set(outs.values())) == 1, outs


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_golden_traces_through_the_address_sanitized_sampler(tmp_path):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("libasan not installed")
    lib = str(tmp_path / "libbsr_stub_asan.so")
    b = subprocess.run(["bash", BUILD, "asan", lib], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stderr[-4000:]
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", BSR_LIB_PATH=lib, BSR_EXEMPT_DISCOVER="1")
    # the GPU box's own trace tests of the C++ sampler, pointed at the sanitized CPU build
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_chain.py"), "-q", "-x",
                        "-m", "gpu", "-p", "no:cacheprovider", "-s", "-k", "native_engine"],
                       env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[:4000]
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_the_sampler_predicts_most_rank_gate_rejections(engine_host_exe):
    """What ends a speculative run of the native sampler is, nine times in ten, a rank-gate rejection nobody predicted
    (codes/funcs.py:1226-1228 draws no accept-uniform behind one: everything generated behind it is void).  The
    sampler predicts them from structure (repeats, linear spans), from an interval estimate of a candidate's scale that
    tracks how close to zero its values come, and from its memory of what the gate has already rejected in the chain's
    current state.  Pinned on the CPU stand-in of the data side (which computes the true ranks): of the gate's
    rejections in 8 chains x 2 000 proposals at least 85 % are predicted, and false alarms stay below the hits' tenth
    (before round 4: 38 % predicted on the GPU box's mix)."""
    exe = engine_host_exe
    r = subprocess.run([exe, "8", "2000"], env=dict(os.environ, BSR_ENGINE_PROF="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    m = re.search(r"events in (\d+) consumed proposals: (\d+) accepts, (\d+) gate rejections nobody predicted, (\d+) predicted "
                  r"rejections that passed the gate; (\d+) of (\d+) gate rejections predicted", r.stderr)
    assert m, r.stderr[-2000:]
    n, acc, missed, false_alarm, hit, rej = (int(v) for v in m.groups())
    assert n == 16000 and rej > 100, m.group(0)
    assert hit >= 0.85 * rej, m.group(0)
    assert false_alarm <= 0.1 * hit, m.group(0)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_memo_answered_accepts_in_groups_of_several_chains(engine_host_exe):
    """an accepted proposal that was answered from the score memo is in no GPU batch; its one-tape
    batch used to take the lane's slot INSIDE the loop over the batch's chains, and every later chain of that batch
    that accepted a GPU-scored proposal then committed by its index into the replaced batch (BSR_E_STATE, or -- index 0
    -- the wrong tape).  Groups of several chains (what `sharded.run_rank` and the bench run), long enough for two
    accepts in one batch: the memo must change nothing -- same digests with BSR_ENGINE_MEMO=1 and 0, whatever the
    grouping, also on the traced single-threaded path (all chains in one batch, commits through `last waited`)."""
    exe = engine_host_exe
    for n_chains in (8, 16):       # (40 000 proposals per chain: 70-125 accepts, a third of them answered from the memo)
        outs = {}
        for groups in ("1", "2", "4"):
            for memo in ("1", "0"):
                r = subprocess.run([exe, str(n_chains), "40000"], env=dict(os.environ, BSR_ENGINE_GROUPS=groups, BSR_ENGINE_MEMO=memo),
                                   capture_output=True, text=True, timeout=900)
                assert r.returncode == 0, (n_chains, groups, memo, r.stderr[-2000:])
                outs[(groups, memo)] = r.stdout
        for memo in ("1", "0"):   # traced: one group, no worker threads, bsr_commit through the last waited batch
            r = subprocess.run([exe, str(n_chains), "40000", "1000" if n_chains == 8 else "0"], env=dict(os.environ, BSR_ENGINE_MEMO=memo),
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, (n_chains, "traced", memo, r.stderr[-2000:])
            outs[("traced", memo)] = r.stdout
        first = outs[("1", "0")]
        assert len(first.splitlines()) == n_chains
        assert all(v == first for v in outs.values()), [k for k, v in outs.items() if v != first]
