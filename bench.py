#!/usr/bin/env python3
"""Headline benchmark: MH proposals scored per second (N=100k, d=10, K=3) on N MI355X + kernel roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--chains C] [--workload c2|c3|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

A "step" is one bsr_score_batch call: B speculative proposals per chain, drawn by the real move mix (all seven
actions, codes/funcs.py:475-480) from a seeded chain state, are scored against the chain's current trees: tree
evaluation over all N rows, rank gate, OLS fit, Gaussian log-likelihood (codes/funcs.py:1212-1235).  X, y and the
chain caches are resident in HBM; the call still uploads the tapes (KBs) and downloads the B result records, because
that is what the C ABI boundary hands over.  Independent chains shard one per rank with no data-path collective
(weak scaling); the only exchange is the RCCL all-gather of the chains' accepted trees after the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))

import numpy as np

WORKLOADS = {
    "c2": dict(N=100_000, d=10, K=3, desc="N=100k, d=10, K=3, 1 chain per GPU (BASELINE configs[1])"),
    "c3": dict(N=100_000, d=10, K=8, desc="N=100k, d=10, K=8, batched multi-proposal (BASELINE configs[2])"),
    "c5": dict(N=1_000_000, d=50, K=3, desc="N=1M, d=50, K=3 (BASELINE configs[4])"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak


def synth(N, d, seed=0):
    """SURVEY 8d recipe: X~U(-3,3), y = 1.35 x0 x1 + 5.5 sin((x0-1)(x1-1)) + 0.1 N(0,1)."""
    rs = np.random.RandomState(seed)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    return X, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="speculative proposals per chain and step")
    ap.add_argument("--chains", type=int, default=1, help="chains per GPU")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--burnin", type=int, default=300, help="real MCMC proposals run before freezing the state")
    ap.add_argument("--cpu-sample", type=float, default=15.0,
                    help="seconds of CPU oracle work for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--depth", type=int, default=4, help="batches in flight, 1..4 (1 = synchronous calls)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from bsr import _lib
    from bsr.chain import Chain, DeviceScorer, run_chains
    from bsr.tape import pack

    W = WORKLOADS[args.workload]
    N, d, K = W["N"], W["d"], W["K"]
    B, C = args.batch, args.chains
    X, y = synth(N, d, seed=0)
    scorer = DeviceScorer(X, y, K, n_chains=C, max_batch=B * C, device=local if world > 1 else 0, dtype=args.dtype)
    ctx = scorer.ctx

    # chain states: seeded chains advanced by a short real run, then frozen
    chains = []
    for c in range(C):
        np.random.seed(1000 + rank * C + c)
        chains.append(Chain(c, scorer, N, d, K, val=10 ** 9))
    run_chains(chains, scorer, batch_per_chain=B, max_props=args.burnin)

    # pre-generate the step inputs: every step scores a fresh batch drawn from the frozen states
    n_batches = args.warmup + args.steps
    # distinct pre-generated batches (the Python sampler needs ~60 ms per batch of 64): long runs cycle through them;
    # every step still stages, uploads and scores its batch in full -- nothing is cached between steps
    n_unique = min(n_batches, max(64, args.warmup + 44))
    packed = []
    feat_counts = []           # |F| of each launch: distinct X columns its tapes reference
    n_nodes = n_trans = 0
    for _ in range(n_unique):
        tapes, chs, ks, sig = [], [], [], []
        for ch in chains:
            for cd in ch.generate(B):
                tapes.append(cd.tape)
                chs.append(ch.index)
                ks.append(cd.k)
                sig.append(cd.new_sigma)
            ch.rng_state = ch._end_state      # keep drawing new proposals from the same frozen state
        rows, off = pack(tapes)
        feats = set()
        for t in tapes:
            n_nodes += len(t)
            n_trans += int(np.isin(t["opcode"], (0, 3, 4, 5)).sum())
            feats.update(int(f) for f in t["feature"][t["opcode"] == 10])
        feat_counts.append(len(feats))
        rec = (rows, off, np.array(chs, np.int32), np.array(ks, np.int32), np.array(sig, np.float64),
               np.zeros(len(tapes), dtype=_lib.SCORE_DTYPE), tapes)
        packed.append(rec + (ctx.prepare(*rec[:5]),))     # input addresses resolved once: the batch is host-resident
    P = len(packed[0][2])

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        r = packed[i % n_unique]
        ctx.score_packed(r[0], r[1], r[2], r[3], r[4], r[5])
    kern_us = np.zeros(5)
    n_timed = 0
    depth = max(1, min(4, args.depth))
    TIMED_EVERY = 4            # HIP events bracket the row pass of every 4th batch of the timed region (two more HIP
                               # calls, ~6 us of host time, on a submission path that is host-bound at B=64)
    barrier()
    t0 = time.perf_counter()
    # several batches in flight: the host stages batch i+1 while the GPU scores batch i (different chain groups in a
    # real run; here every batch is drawn from frozen chain states, so there is no dependency between batches)
    tickets = []
    for i in range(args.warmup, n_batches):
        r = packed[i % n_unique]
        timed = (i - args.warmup) % TIMED_EVERY == 0
        if timed:
            ctx.set_profiling(1)
        tickets.append((ctx.score_submit_prepared(r[7]), r, timed))
        if timed:
            ctx.set_profiling(0)
        if len(tickets) >= depth:
            t, rr, tm = tickets.pop(0)
            ctx.score_wait(t, rr[5])
            if tm:
                kern_us += ctx.last_timing()
                n_timed += 1
    while tickets:
        t, rr, tm = tickets.pop(0)
        ctx.score_wait(t, rr[5])
        if tm:
            kern_us += ctx.last_timing()
            n_timed += 1
    barrier()
    elapsed = time.perf_counter() - t0
    kern_us /= max(1, n_timed)
    # the same kernel with nothing else on the GPU (one batch at a time): with several batches in flight the events of
    # the timed region also span time the kernel shares the chip with the other streams' small kernels, and
    # rocprofv3 serialises dispatches, so this is the figure its kernel stats reproduce
    n_iso = min(args.steps, 50)
    kern_iso = 0.0
    ctx.set_profiling(1)
    for i in range(args.warmup, args.warmup + n_iso):
        r = packed[i % n_unique]
        ctx.score_packed(r[0], r[1], r[2], r[3], r[4], r[5])
        kern_iso += ctx.last_timing()[0]
    kern_iso /= n_iso
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the one exchange of the path: gather every chain's current (accepted) trees over RCCL (outside the timed region)
    gathered = None
    gather_via = None
    if dist is not None:
        import torch
        from bsr.dist import pack_chain_record, RECORD_BYTES, TorchGather
        rec = np.concatenate([pack_chain_record(ch) for ch in chains])
        try:     # RCCL communicator owned by the C ABI (bsr_comm_*), unique id handed out through the launcher's group
            uid = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.from_numpy(ctx.comm_unique_id()))
            dist.broadcast(uid, 0)
            ctx.comm_init(world, rank, uid.cpu().numpy())
            gathered = ctx.comm_allgather(rec)
            gather_via = "bsr_comm_allgather (RCCL via C ABI)"
        except Exception as exc:   # keep the bench line: fall back to the launcher's own RCCL group
            sys.stderr.write("C-ABI gather failed (%r); using torch.distributed all_gather\n" % (exc,))
            gathered = TorchGather(device="cuda").allgather(rec)
            gather_via = "torch.distributed all_gather (RCCL)"
        assert gathered.shape == (world, RECORD_BYTES * C)

    if rank == 0:
        total_props = world * P * args.steps
        value = total_props / elapsed
        # roofline of the dominant kernel (tree-eval + projection pass), SURVEY 8d formula:
        # bytes = s * N * (|F| + 1 + C_r + C_w): features referenced, y, K-1 cached sibling columns per chain,
        # C_w = 0 (candidate columns are scratch; reported separately)
        s = 8 if args.dtype == "f64" else 4
        n_feat = float(np.mean(feat_counts))          # per launch, averaged over the batches of the run
        alg_bytes = int(s * N * (n_feat + 1 + C * (K - 1)))
        p1 = kern_iso * 1e-6
        out = {
            "metric": "MH proposals scored/sec (N=100k,d=10,K=3) at 1/2/4/8 MI355X; HBM GB/s",
            "value": value, "unit": "proposals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": W["desc"], "N": N, "d": d, "K": K, "chains_per_gpu": C,
                       "proposals_per_step_per_gpu": P, "speculative_batch": B, "parallelism": "chains x%d" % world,
                       "avg_nodes_per_tape": n_nodes / (n_unique * P),
                       "transcendental_node_frac": n_trans / max(1, n_nodes)},
            "roofline": {"bound": "hbm", "achieved": alg_bytes / p1 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / p1 / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_rows<PROJECT> (tree-eval + projection)", "kernel_us": kern_iso,
                         "kernel_us_in_timed_region": kern_us[0],
                         "algorithmic_bytes": alg_bytes, "features_per_launch": n_feat},
            "batches_in_flight": depth,
        }
        # HBM traffic per launch of that kernel from the committed rocprofv3 PMC run of this same command
        # (tools/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH x2 on gfx950)
        import glob
        tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_%s_B%d.json" % (args.workload, B))))
        tpath = tpaths[-1] if tpaths else ""          # the latest committed run of this workload
        if C == 1 and args.dtype == "f64" and tpath:
            for kname, rec in json.load(open(tpath)).items():
                if "k_rows" in kname:
                    out["roofline"]["traffic"] = rec["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = os.path.relpath(tpath, ROOT)
        if args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(X, y, K, chains, packed[args.warmup:], args.cpu_sample)
        if gathered is not None:
            out["gathered_records"] = int(gathered.shape[0] * C)
            out["gather"] = gather_via
        print(json.dumps(out))
    scorer.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(X, y, K, chains, batches, budget_s):
    """Times the oracle's reference-faithful restatement of the same scoring work on the host CPU (1 thread):
    per proposal K+1 tree evaluations with per-element exp/inv loops, SVD rank gate, two ylogLike passes
    (codes/funcs.py:1212-1235).  Bounded sample: proposals of the timed batches, in order, for ~budget_s seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pandas as pd
    import bsr_oracle as O
    from bsr.tape import unflatten

    def onode(n):
        m = O.ONode(n.depth)
        m.type, m.operator, m.op_ind, m.feature, m.a, m.b = n.type, n.operator, n.op_ind, n.feature, n.a, n.b
        m.left = onode(n.left) if n.left is not None else None
        m.right = onode(n.right) if n.right is not None else None
        return m
    Xdf = pd.DataFrame(X)
    ys = pd.Series(y)
    t0 = time.perf_counter()
    done = 0
    for batch in batches:
        tapes, chs, ks, sig = batch[6], batch[2], batch[3], batch[4]
        for i in range(len(tapes)):
            ch = chains[int(chs[i])]
            k = int(ks[i])
            new_o = np.zeros((len(y), K))
            old_o = np.zeros((len(y), K))
            with np.errstate(all="ignore"):
                for j in range(K):
                    if j == k:
                        new_o[:, j] = O.allcal(onode(unflatten(tapes[i])), Xdf, faithful=True)[:, 0]
                        old_o[:, j] = O.allcal(onode(ch.roots[j]), Xdf, faithful=True)[:, 0]
                    else:
                        col = O.allcal(onode(ch.roots[j]), Xdf, faithful=True)[:, 0]
                        new_o[:, j] = col
                        old_o[:, j] = col
                try:
                    full = np.linalg.matrix_rank(new_o) == K
                except np.linalg.LinAlgError:
                    full = False
                if full:
                    O.yloglike(ys, new_o, float(sig[i]))
                    O.yloglike(ys, old_o, ch.sigma)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "proposals/s", "cores": 1, "kind": "port",
           "sample": "%d proposals of the timed batches in order, oracle reference-faithful flavour "
                     "(K+1 allcal with per-element exp/inv loops + matrix_rank + 2 ylogLike), %.1f s" % (done, dt)}
    # second flavour: the same CPU path written the way a numpy user would -- vectorised exp/inv, the
    # sibling columns and the old log-likelihood cached per chain -- so the ratio is not only "a Python loop removed"
    t1 = time.perf_counter()
    fair = 0
    cache = {}
    budget2 = budget_s / 3.0
    for batch in batches:
        tapes, chs, ks, sig = batch[6], batch[2], batch[3], batch[4]
        for i in range(len(tapes)):
            ch = chains[int(chs[i])]
            k = int(ks[i])
            with np.errstate(all="ignore"):
                if ch.index not in cache:
                    cols = np.stack([O.allcal(onode(ch.roots[j]), Xdf, faithful=False)[:, 0] for j in range(K)], axis=1)
                    cache[ch.index] = (cols, O.yloglike(ys, cols, ch.sigma) if np.all(np.isfinite(cols)) else None)
                new_o = cache[ch.index][0].copy()
                new_o[:, k] = O.allcal(onode(unflatten(tapes[i])), Xdf, faithful=False)[:, 0]
                try:
                    full = np.linalg.matrix_rank(new_o) == K
                except np.linalg.LinAlgError:
                    full = False
                if full:
                    O.yloglike(ys, new_o, float(sig[i]))
            fair += 1
            if time.perf_counter() - t1 > budget2:
                break
        if time.perf_counter() - t1 > budget2:
            break
    dt2 = time.perf_counter() - t1
    out["vectorised"] = {"value": fair / dt2, "unit": "proposals/s", "cores": 1,
                         "sample": "%d proposals, vectorised exp/inv, cached sibling columns and old log-likelihood, %.1f s"
                                   % (fair, dt2)}
    return out


if __name__ == "__main__":
    main()
