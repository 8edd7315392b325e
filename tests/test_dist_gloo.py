"""world_size-2 CPU test (gloo) of the multi-GPU path: chain sharding and the accepted-tree gather."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, os.path.join(%(root)r, "mcmc-symreg_amd"))
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    sys.path.insert(0, os.path.join(%(root)r, "oracle"))
    import numpy as np
    import torch.distributed as dist
    from bsr import dist as D
    from bsr.chain import Chain, run_chains
    from bsr.node import Express
    from test_host_driver import OracleScorer

    class TorchGather:
        """All-gather through torch.distributed (gloo on CPU): test scaffolding only -- the product's gather is RCCL through
        the C ABI (bsr.dist.RcclGather), and nothing under mcmc-symreg_amd/ imports torch."""

        def world(self):
            return dist.get_world_size()

        def allgather(self, send):
            import torch
            t = torch.from_numpy(np.ascontiguousarray(send, dtype=np.uint8))
            outs = [torch.empty_like(t) for _ in range(self.world())]
            dist.all_gather(outs, t)
            return np.stack([o.numpy() for o in outs])

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rs = np.random.RandomState(0)
    X = rs.uniform(-3, 3, size=(60, 2))
    y = X[:, 0] * X[:, 1] + np.sin(X[:, 0])
    n_chains = 5
    mine = D.shard(n_chains, world, rank)
    per_rank = (n_chains + world - 1) // world
    recs = []
    for c in mine:
        sc = OracleScorer(X, y, 2)
        np.random.seed(1000 + c)
        ch = Chain(0, sc, len(y), 2, 2, val=15)
        run_chains([ch], sc, batch_per_chain=8)
        recs.append(D.pack_record(c, ch.roots, ch.Beta, ch.sigma, ch.errs, ch.n_props, ch.n_accept))
    allrecs = D.gather_chains(TorchGather(), recs, per_rank)
    assert [r["chain"] for r in allrecs] == list(range(n_chains)), [r["chain"] for r in allrecs]
    lines = ["%%d|%%d|%%s|%%r" %% (r["chain"], r["n_props"], ";".join(Express(t) for t in r["roots"]),
                               [round(float(v), 10) for v in r["beta"].reshape(-1)]) for r in allrecs]
    open(os.path.join(%(out)r, "rank%%d.txt" %% rank), "w").write("\\n".join(lines))
    dist.barrier()
    dist.destroy_process_group()
''')


def _run_world(world, out, port):
    script = os.path.join(out, "worker.py")
    open(script, "w").write(WORKER % {"root": ROOT, "out": out})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [open(os.path.join(out, "rank%d.txt" % k)).read() for k in range(world)]


def test_sharded_chains_gather_is_world_size_invariant(tmp_path):
    d2 = tmp_path / "w2"
    d1 = tmp_path / "w1"
    d2.mkdir()
    d1.mkdir()
    two = _run_world(2, str(d2), 29541)
    one = _run_world(1, str(d1), 29542)
    assert two[0] == two[1]            # every rank holds the full, ordered result
    assert two[0] == one[0]            # and it does not depend on how chains were sharded
    assert len(two[0].splitlines()) == 5


def test_record_roundtrip():
    sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
    from bsr import dist as D
    from bsr.node import Express
    from conftest import load_golden, node_from_spec
    g = load_golden("g2_grow.json")
    roots = [node_from_spec(c["tree"]) for c in g["cases"][:3]]
    rec = D.pack_record(7, roots, np.arange(4.0), 0.5, [1.0, 0.7], 123, 2)
    assert rec.size == D.record_bytes(rec)
    u = D.unpack_record(rec)
    assert u["chain"] == 7 and u["K"] == 3 and u["n_props"] == 123 and u["n_accept"] == 2
    assert [Express(t) for t in u["roots"]] == [Express(t) for t in roots]
    assert np.array_equal(u["beta"].reshape(-1), np.arange(4.0)) and u["best_rmse"] == 0.7
    assert D.shard(7, 3, 1) == [1, 4]


def test_a_corrupt_or_short_record_raises_instead_of_truncating():
    """unpack_record checked its variable-length header with asserts only (gone under `python -O`; slices of
    a short buffer truncate silently).  Every way a record can disagree with its buffer raises ValueError."""
    sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
    import pytest
    from bsr import dist as D
    from conftest import load_golden, node_from_spec
    g = load_golden("g2_grow.json")
    roots = [node_from_spec(c["tree"]) for c in g["cases"][:3]]
    rec = D.pack_record(7, roots, np.arange(4.0), 0.5, [1.0, 0.7], 123, 2)
    with pytest.raises(ValueError):
        D.unpack_record(rec[:-8])                       # short buffer
    with pytest.raises(ValueError):
        D.unpack_record(rec[:D.HEADER_BYTES - 8])       # not even a header
    for word, value in ((1, D.MAX_K + 1), (1, -1), (4, -3), (4, 5), (16, -1), (16, 10 ** 6), (9, rec.size + 8)):
        bad = rec.copy()
        bad[:D.HEADER_I32 * 4].view(np.int32)[word] = value
        with pytest.raises(ValueError):
            D.unpack_record(bad)
    assert D.unpack_record(np.concatenate([rec, rec]))["chain"] == 7   # a longer buffer is fine: the header bounds the record
