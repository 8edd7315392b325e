"""Config 4: the reference's independent restarts (codes/bsr_class.py:99, 270-276) sharded over
the GPUs of one node, one process per GPU, native sampler per rank, one RCCL all-gather of the chains' outcomes.

Chain c runs on rank c % world, seeded like np.random.seed(seeds[c]) immediately before the chain
(codes/bsr_class.py:99-163 then draws from that stream), so a chain's result does not depend on the world size or on
which other chains share its launches.  There is no data-path collective; X and y are uploaded by every rank.

  python -m bsr.sharded --data DIR --out FILE [...]      one rank (started by fit_sharded / an external launcher)
  fit_sharded(X, y, devices=[0, 1, ...], ...)            parent side: spawns the ranks, returns the gathered chains
"""
import argparse
import os
import sys
import tempfile

import numpy as np

from . import dist as D


def run_rank(X, y, K, seeds, rank=0, world=1, device=0, batch=None, val=100, beta=-1, chains_per_launch=8,
             dtype="f64", y_is_series=True, max_props=-1, scorer=None, ops=None, op_weights=None):
    """Runs this rank's share of len(seeds) chains on `device` with the native sampler and gathers every rank's
    records.  Returns (raw records of ALL chains ordered by chain id: a list of uint8 arrays, decode with
    bsr.dist.unpack_record; this rank's counters).

    The rendezvous and the communicator come FIRST: a rank is "up" (bsr.launch.spawn's init_timeout) once it has
    joined the node's gather, not once its chains are done -- a fit may take hours, joining may not."""
    from .chain import DeviceScorer
    from .native import NativeEngine
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
    n_chains = len(seeds)
    if not batch:
        from .native import default_batch
        batch = default_batch(X.shape[0], X.shape[1], K)
    mine = D.shard(n_chains, world, rank)
    n_slots = max(1, min(chains_per_launch, len(mine)))
    own = scorer is None
    if own:
        from .native import batch_shape
        tc, tb = batch_shape(n_slots, batch, K)
        scorer = DeviceScorer(X, y, K, n_chains=n_slots, max_batch=max(4, batch * n_slots), device=device, dtype=dtype,
                              typical_chains=tc, typical_batch=tb)
    recs = []
    stats = {"proposals": 0, "accepts": 0, "rank_rejects": 0, "discarded": 0, "chains": len(mine)}
    try:
        gather, rdv = D.connect(scorer.ctx, rank, world)
        eng = NativeEngine(scorer.ctx, n_slots, X.shape[1], beta=beta, val=val, y_is_series=y_is_series)
        if ops is not None:
            eng.set_ops(ops, op_weights if op_weights is not None else [1.0 / len(ops)] * len(ops))
        try:
            todo = list(mine)
            while todo:
                wave, todo = todo[:n_slots], todo[n_slots:]
                for slot, c in enumerate(wave):
                    eng.seed(slot, seeds[c])
                    eng.init_chain(slot)
                eng.run(batch_per_chain=batch, max_props=max_props)
                for slot, c in enumerate(wave):
                    r = eng.result(slot)
                    recs.append(D.pack_record(c, None, r["beta"], r["sigma"], r["errs"], r["n_props"], r["n_accept"],
                                              r["n_rank_rejects"], r["n_discarded"], tapes_in=r["tapes"]))
                    stats["proposals"] += r["n_props"]
                    stats["accepts"] += r["n_accept"]
                    stats["rank_rejects"] += r["n_rank_rejects"]
                    stats["discarded"] += r["n_discarded"]
        finally:
            eng.close()
        allrecs = D.gather_raw(gather, recs)
    finally:
        if own:
            scorer.close()
    return allrecs, stats


def fit_sharded(X, y, K=3, seeds=None, n_chains=None, devices=(0,), batch=None, val=100, beta=-1,
                chains_per_launch=8, dtype="f64", y_is_series=True, timeout=None, env_extra=None, ops=None,
                op_weights=None):
    """Parent side.  Spawns one fresh process per entry of `devices` (this process needs no GPU and must not be
    asked to share its own context), waits, and returns the gathered chain records ordered by chain id."""
    from .launch import spawn
    if seeds is None:
        seeds = [1000 + c for c in range(int(n_chains))]
    work = tempfile.mkdtemp(prefix="bsr_sharded_")
    try:
        np.save(os.path.join(work, "X.npy"), np.ascontiguousarray(np.asarray(X, dtype=np.float64)))
        np.save(os.path.join(work, "y.npy"), np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1)))
        np.save(os.path.join(work, "seeds.npy"), np.asarray(seeds, dtype=np.int64))
        out = os.path.join(work, "gathered.npz")
        pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = {"PYTHONPATH": pkg_parent + os.pathsep + os.environ.get("PYTHONPATH", ""),
               "BSR_DEVICES": ",".join(str(int(d)) for d in devices)}
        if env_extra:
            env.update(env_extra)
        argv = ["-m", "bsr.sharded", "--data", work, "--out", out, "--K", str(K), "--batch", str(int(batch or 0)),
                "--val", str(val), "--beta", repr(float(beta)), "--chains-per-launch", str(chains_per_launch),
                "--dtype", dtype, "--y-is-series", "1" if y_is_series else "0"]
        if ops is not None:
            w = op_weights if op_weights is not None else [1.0 / len(ops)] * len(ops)
            argv += ["--ops", ",".join(ops), "--op-weights", ",".join(repr(float(v)) for v in w)]
        # no limit on the fit itself unless the caller (or BSR_SPAWN_TIMEOUT) sets one; the start-up -- rendezvous and
        # ncclCommInitRank, the one place ranks can hang without dying -- is bounded on its own
        codes, _ = spawn(len(devices), argv, env_extra=env, timeout=timeout, relay_rank0_stdout=False,
                         init_timeout=float(os.environ.get("BSR_INIT_TIMEOUT", "900")) if len(devices) > 1 else None)
        if getattr(codes, "timed_out", None):
            raise TimeoutError("bsr.sharded: %s (rank exit codes %r)" % (codes.reason, list(codes)))
        if any(c != 0 for c in codes):
            raise RuntimeError("bsr.sharded: rank exit codes %r" % (list(codes),))
        return [D.unpack_record(r) for r in D.load_records(out)]
    finally:
        for name in os.listdir(work):
            try:
                os.unlink(os.path.join(work, name))
            except OSError:
                pass
        try:
            os.rmdir(work)
        except OSError:
            pass


def main(argv=None):
    ap = argparse.ArgumentParser(description="one rank of a sharded BSR fit (see fit_sharded)")
    ap.add_argument("--data", required=True, help="directory holding X.npy, y.npy, seeds.npy")
    ap.add_argument("--out", required=True, help="rank 0 writes the gathered records here (.npz: bsr.dist.load_records)")
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="speculative proposals per chain and batch (0: bsr.native.default_batch)")
    ap.add_argument("--val", type=int, default=100)
    ap.add_argument("--beta", type=float, default=-1.0)
    ap.add_argument("--chains-per-launch", type=int, default=8)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--y-is-series", type=int, default=1)
    ap.add_argument("--ops", default="", help="comma-separated operator names (default: the reference's ten)")
    ap.add_argument("--op-weights", default="", help="comma-separated prior weights, one per operator")
    args = ap.parse_args(argv)
    from .launch import rank_env
    rank, world, local = rank_env()
    devs = [int(v) for v in os.environ.get("BSR_DEVICES", "").split(",") if v != ""]
    device = devs[local] if local < len(devs) else D.local_device(local)
    X = np.load(os.path.join(args.data, "X.npy"))
    y = np.load(os.path.join(args.data, "y.npy"))
    seeds = [int(s) for s in np.load(os.path.join(args.data, "seeds.npy"))]
    allrecs, stats = run_rank(X, y, args.K, seeds, rank=rank, world=world, device=device, batch=args.batch,
                              val=args.val, beta=args.beta, chains_per_launch=args.chains_per_launch,
                              dtype=args.dtype, y_is_series=bool(args.y_is_series),
                              ops=args.ops.split(",") if args.ops else None,
                              op_weights=[float(v) for v in args.op_weights.split(",")] if args.op_weights else None)
    if rank == 0:
        tmp = args.out + ".tmp"
        D.save_records(tmp, allrecs)
        os.replace(tmp, args.out)
    sys.stderr.write("bsr.sharded rank %d/%d on device %d: %d chains, %d proposals, %d accepts\n"
                     % (rank, world, device, stats["chains"], stats["proposals"], stats["accepts"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
