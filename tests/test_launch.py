"""Torch-free multi-process plumbing of the multi-GPU path, exercised on CPU: the child-process
launcher, the rendezvous directory that carries the RCCL unique id, the record gather on top of an all-gather, and
bench.py's refusal to run when --gpus disagrees with the launcher."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mcmc-symreg_amd")

WORKER = textwrap.dedent('''
    import hashlib, os, sys
    sys.path.insert(0, %(pkg)r)
    import numpy as np
    from bsr import dist as D
    from bsr.launch import Rendezvous, rank_env
    rank, world, local = rank_env()
    assert local == rank
    rdv = Rendezvous(rank, world)
    uid = rdv.broadcast("uid", lambda: os.urandom(128), nbytes=128)       # what bsr_comm_unique_id would hand out
    g = D.FileGather(rdv)
    got = g.allgather(np.frombuffer(uid, dtype=np.uint8))
    assert got.shape == (world, 128) and all(np.array_equal(got[r], got[0]) for r in range(world))
    assert D.allreduce_max(g, float(rank) + 0.5) == world - 0.5
    D.barrier(g)
    # the chain-record gather on top of it: 5 chains dealt round-robin, every rank ends with all 5 in chain order
    n_chains = 5
    recs = []
    for c in D.shard(n_chains, world, rank):
        tape = np.zeros(1, dtype=D.NODE_DTYPE)
        tape[0] = (10, -1, -1, c %% 3, 0.0, 0.0)
        recs.append(D.pack_record(c, None, np.arange(3.0) + c, 0.5 + c, [1.0 / (c + 1)] * c, 100 + c, c, 7, 9,
                                  tapes_in=[tape, tape]))
    raw = D.gather_raw(g, recs, (n_chains + world - 1) // world)
    out = [D.unpack_record(raw[i]) for i in range(raw.shape[0])]
    assert [r["chain"] for r in out] == list(range(n_chains))
    for c, r in enumerate(out):
        assert r["n_props"] == 100 + c and r["n_accept"] == c and r["n_rank_rejects"] == 7 and r["n_discarded"] == 9
        assert r["errs"] == [1.0 / (c + 1)] * c and r["sigma"] == 0.5 + c
        assert np.array_equal(r["beta"].reshape(-1), np.arange(3.0) + c)
        assert int(r["tapes"][0]["feature"][0]) == c %% 3
    if rank == 0:
        print("DIGEST", hashlib.sha1(raw.tobytes()).hexdigest())
''')


def _run(world, tmp_path):
    script = tmp_path / ("worker%d.py" % world)
    script.write_text(WORKER % {"pkg": PKG})
    code = ("import sys; sys.path.insert(0, %r); from bsr.launch import spawn; "
            "codes, text = spawn(%d, [%r], relay_rank0_stdout=False, timeout=120); print(codes); print(text)"
            % (PKG, world, str(script)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0] == str([0] * world), r.stdout + r.stderr[-4000:]
    return [ln for ln in lines if ln.startswith("DIGEST")][0]


def test_spawned_ranks_meet_and_gather_world_size_invariant(tmp_path):
    three = _run(3, tmp_path)
    one = _run(1, tmp_path)
    assert three == one          # the gathered, ordered records do not depend on how the chains were sharded


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 1 but the launcher started 2" in r.stderr


def test_record_keeps_the_rmse_history_and_counters():
    sys.path.insert(0, PKG)
    from bsr import dist as D
    tape = np.zeros(3, dtype=D.NODE_DTYPE)
    tape[0] = (10, -1, -1, 1, 0.0, 0.0)
    tape[1] = (10, -1, -1, 0, 0.0, 0.0)
    tape[2] = (9, 0, 1, 0, 0.0, 0.0)
    errs = list(np.linspace(2.0, 1.0, 1500))                 # longer than the record keeps: the tail survives
    rec = D.pack_record(3, None, [0.1, 0.2], 0.7, errs, 5000, 1500, 11, 22, tapes_in=[tape])
    u = D.unpack_record(rec)
    assert u["n_errs"] == 1500 and len(u["errs"]) == D.ERRS_CAP and u["errs"][-1] == 1.0
    assert u["errs"] == [float(v) for v in errs[-D.ERRS_CAP:]]
    assert (u["n_rank_rejects"], u["n_discarded"], u["K"]) == (11, 22, 1)
    from bsr.node import Express
    assert Express(u["roots"][0]) == "(x[1])*(x[0])"
