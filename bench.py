#!/usr/bin/env python3
"""Headline benchmark: MH proposals scored per second (N=100k, d=10, K=3) on N MI355X + kernel roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--chains C] [--workload c2|c3|c4|c5]

With --gpus N > 1 and no launcher variables in the environment, this process starts N fresh child processes (one per
GPU, RANK/LOCAL_RANK/WORLD_SIZE set) before anything touches the GPU, relays rank 0's JSON line and exits with the
worst child exit code.  Under an external launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`) each process is one rank already.  Ranks meet through a rendezvous directory for the 128-byte RCCL unique
id; barrier, max-over-ranks timing and the gather of accepted trees are RCCL all-gathers through the C ABI
(bsr_comm_allgather).  No PyTorch anywhere.

A "step" is one bsr_score_batch call: B speculative proposals per chain, drawn by the real move mix (all seven
actions, codes/funcs.py:475-480) from a seeded chain state, are scored against the chain's current trees: tree
evaluation over all N rows, rank gate, OLS fit, Gaussian log-likelihood (codes/funcs.py:1212-1235).  X, y and the
chain caches are resident in HBM; the call still uploads the tapes (KBs) and downloads the B result records, because
that is what the C ABI boundary hands over.  Independent chains shard over the ranks with no data-path collective
(weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
# this process is the benchmark and nothing else: confine it (caller included) to one L3 domain of the host.  The
# library by itself only places its own threads (DESIGN 7 "CPU placement"); the setting is reported under env_knobs.
os.environ.setdefault("BSR_PIN", "1")

WORKLOADS = {
    "c2": dict(N=100_000, d=10, K=3, chains=1, batch=64,
               desc="N=100k, d=10, K=3, 1 chain per GPU (BASELINE configs[1])"),
    "c3": dict(N=100_000, d=10, K=8, chains=1, batch=64,
               desc="N=100k, d=10, K=8, batched multi-proposal (BASELINE configs[2])"),
    "c4": dict(N=100_000, d=10, K=3, chains=8, batch=32,
               desc="N=100k, d=10, K=3, 8 chains per GPU (per-GPU share of BASELINE configs[3])"),
    "c5": dict(N=1_000_000, d=50, K=3, chains=1, batch=64,
               desc="N=1M, d=50, K=3 (BASELINE configs[4])"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
SETTLE_S = 1.0        # untimed pipelined steps in front of every timed region, seconds (scaled down with --min-time)
MIN_TIMED_S = 1.0       # the timed region is repeated (whole multiples of --steps) until it lasts this long
N_REGIONS = 3           # timed regions per leg: the median one is reported, all of them listed


def synth(N, d, seed=0):
    """SURVEY 8d recipe: X~U(-3,3), y = 1.35 x0 x1 + 5.5 sin((x0-1)(x1-1)) + 0.1 N(0,1)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    X = rs.uniform(-3, 3, size=(N, d))
    y = 1.35 * X[:, 0] * X[:, 1] + 5.5 * np.sin((X[:, 0] - 1) * (X[:, 1] - 1)) + 0.1 * rs.standard_normal(N)
    return X, y


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0, help="speculative proposals per chain and step (0: workload default)")
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (0: workload default)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--burnin", type=int, default=300, help="real MCMC proposals run before freezing the state")
    ap.add_argument("--cpu-sample", type=float, default=15.0,
                    help="seconds of CPU oracle work for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--depth", type=int, default=0,
                    help="batches in flight, 1..8 (1 = synchronous calls); 0: 8 where K <= 4, 6 above.  The GPU runs four "
                         "chains of dependent kernels at a time (DESIGN 7.1) and the batches behind them wait in the queues, "
                         "so that a chain's end is followed by the next one's start without a host round trip; with the "
                         "long k_solve of K >= 5 more than six only wait longer (C2: 8 measures 1 %% above 6; C3: 6 %% below)")
    ap.add_argument("--extras", type=int, default=-1,
                    help="1: also run short legs of the other configs (c3, c5, c4's per-GPU share, native-engine chain "
                         "throughput) and report them under 'extra'; 0: headline only; -1: on for the default workload")
    ap.add_argument("--min-time", type=float, default=MIN_TIMED_S)
    ap.add_argument("--rows", type=int, default=0,
                    help="diagnostic only: override the workload's N (e.g. 2000 makes the GPU work negligible, so "
                         "ms_per_step shows the host-side cost of a step); the line is then not a benchmark result")
    return ap.parse_args()


def launch_children(args):
    """--gpus N without a launcher: N fresh child processes, one per device.  Nothing here touches HIP."""
    from bsr.launch import spawn
    argv = [os.path.abspath(__file__)] + sys.argv[1:]
    # the run itself is as long as it is; the start-up (rendezvous, ncclCommInitRank) is what can hang: bounded alone
    codes, text = spawn(args.gpus, argv, init_timeout=float(os.environ.get("BSR_INIT_TIMEOUT", "900")))
    bad = [c for c in codes if c != 0]
    if bad:
        sys.stderr.write("bench.py: rank exit codes %r%s\n" % (list(codes), (" -- " + codes.reason) if codes.reason else ""))
        return max(abs(c) for c in bad) or 1
    return 0


class Ranks:
    """This rank's view of the job: device, gather (RCCL all-gather through the C ABI) and the helpers built on it."""

    def __init__(self):
        from bsr.launch import rank_env
        self.rank, self.world, self.local = rank_env()
        self.gather = None
        self.via = None

    def device(self):
        if os.environ.get("BSR_SHARE_DEVICE") == "1":
            return 0
        if self.world <= 1:
            return 0
        from bsr.dist import local_device          # (a launcher may have narrowed the visible devices to this rank's own)
        return local_device(self.local)

    def connect(self):
        """One small context per rank that lives as long as the process and owns the RCCL communicator."""
        import numpy as np
        from bsr import dist as D
        from bsr.device import DeviceContext
        self.comm_ctx = DeviceContext(np.zeros((1, 1)), None, K=0, n_chains=0, max_batch=1, device=self.device())
        self.gather, _ = D.connect(self.comm_ctx, self.rank, self.world)
        self.via = {"RcclGather": "bsr_comm_allgather (RCCL via C ABI)", "SoloGather": None,
                    "FileGather": "rendezvous directory (BSR_SHARE_DEVICE=1 test mode: ranks share one GPU)"}[
                        type(self.gather).__name__]

    def barrier(self):
        from bsr import dist as D
        if self.world > 1:
            D.barrier(self.gather)

    def max(self, v):
        from bsr import dist as D
        return D.allreduce_max(self.gather, v) if self.world > 1 else v


def build_workload(name, args, ranks, n_unique_min=64):
    """Uploads the data, burns the chains in and pre-generates the step inputs of one workload."""
    import numpy as np
    from bsr import _lib
    from bsr.chain import Chain, DeviceScorer, run_chains
    from bsr.tape import pack
    W = WORKLOADS[name]
    N, d, K = W["N"], W["d"], W["K"]
    if getattr(args, "rows", 0):
        N = args.rows
    B = args.batch or W["batch"]
    C = args.chains or W["chains"]
    X, y = synth(N, d, seed=0)
    scorer = DeviceScorer(X, y, K, n_chains=C, max_batch=B * C, device=ranks.device(), dtype=args.dtype)
    ctx = scorer.ctx
    chains = []
    for c in range(C):
        np.random.seed(1000 + ranks.rank * C + c)
        chains.append(Chain(c, scorer, N, d, K, val=10 ** 9))
    run_chains(chains, scorer, batch_per_chain=B, max_props=args.burnin)
    return dict(name=name, W=W, N=N, d=d, K=K, B=B, C=C, X=X, y=y, scorer=scorer, ctx=ctx, chains=chains,
                pack=pack, lib=_lib)


def generate_batches(wl, n_unique):
    """Distinct pre-generated batches (the Python sampler needs ~60 ms per batch of 64): long runs cycle through them;
    every step still stages, uploads and scores its batch in full -- nothing is cached between steps."""
    import numpy as np
    packed, feat_counts = [], []
    n_nodes = n_trans = 0
    B, ctx, _lib = wl["B"], wl["ctx"], wl["lib"]
    for _ in range(n_unique):
        tapes, chs, ks, sig = [], [], [], []
        for ch in wl["chains"]:
            for cd in ch.generate(B):
                tapes.append(cd.tape)
                chs.append(ch.index)
                ks.append(cd.k)
                sig.append(cd.new_sigma)
            ch.rng_state = ch._end_state      # keep drawing new proposals from the same frozen state
        rows, off = wl["pack"](tapes)
        feats = set()
        for t in tapes:
            n_nodes += len(t)
            n_trans += int(np.isin(t["opcode"], (0, 3, 4, 5)).sum())
            feats.update(int(f) for f in t["feature"][t["opcode"] == 10])
        feat_counts.append(len(feats))
        rec = (rows, off, np.array(chs, np.int32), np.array(ks, np.int32), np.array(sig, np.float64),
               np.zeros(len(tapes), dtype=_lib.SCORE_DTYPE), tapes)
        packed.append(rec + (ctx.prepare(*rec[:5]), rec[5].ctypes.data))   # addresses resolved once: the batch is host-resident
    wl.update(packed=packed, feat_counts=feat_counts, n_nodes=n_nodes, n_trans=n_trans, P=len(packed[0][2]))
    return wl


def pick_depth(depth, K):
    return depth if depth > 0 else (8 if K <= 4 else 6)


def timed_region(wl, ranks, steps, warmup, depth, min_time):
    """W warm-up steps, then R x `steps` timed steps between barriers (R chosen so the region lasts >= min_time),
    several batches in flight; HIP events bracket the row pass of every 16th batch on the stream it runs on."""
    import numpy as np
    ctx, packed = wl["ctx"], wl["packed"]
    n_unique = len(packed)
    for i in range(warmup):
        r = packed[i % n_unique]
        ctx.score_packed(r[0], r[1], r[2], r[3], r[4], r[5])
    kern_us = np.zeros(5)
    n_timed = [0]
    depth = max(1, min(8, pick_depth(depth, wl["K"])))
    TIMED_EVERY = 16   # (a timed batch costs the caller three extra calls and the stream two event records)

    def run_steps(n, first):
        """n pipelined steps: the host stages batch i+1 while the GPU scores batch i (different chain groups in a real
        run; here every batch is drawn from frozen chain states, so batches do not depend on each other)."""
        tickets = []
        for i in range(n):
            r = packed[(first + i) % n_unique]
            timed = i % TIMED_EVERY == 0
            if timed:
                ctx.set_profiling(1)
            tickets.append((ctx.score_submit_prepared(r[7]), r, timed))
            if timed:
                ctx.set_profiling(0)
            if len(tickets) >= depth:
                t, rr, tm = tickets.pop(0)
                ctx.score_wait_ptr(t, rr[8])
                if tm:
                    kern_us[:] += ctx.last_timing()
                    n_timed[0] += 1
        while tickets:
            t, rr, tm = tickets.pop(0)
            ctx.score_wait_ptr(t, rr[8])
            if tm:
                kern_us[:] += ctx.last_timing()
                n_timed[0] += 1

    repeats = 1
    if min_time > 0:
        # untimed probe of the pipelined step time (more warm-up, in effect) sizes the timed region
        n_probe = max(4, min(steps, 40))
        t_w = time.perf_counter()
        run_steps(n_probe, warmup)
        per_step = (time.perf_counter() - t_w) / n_probe
        # ... and a settling phase, untimed like the warm-up: one of the round's nine default runs timed its first leg
        # while the process was still settling (the probe above saw 18.8 us per step, the timed second averaged 14.8, every
        # later leg of the same process ran at the usual rate: profiles/r04z_bench_default_noisy.json).  SETTLE_S of
        # pipelined steps first, then the probe again.
        if SETTLE_S > 0:
            n_settle = int(min(20000, max(n_probe, SETTLE_S * min(1.0, min_time) / max(1e-7, per_step))))
            run_steps(n_settle, warmup)
            t_w = time.perf_counter()
            run_steps(n_probe, warmup)
            per_step = (time.perf_counter() - t_w) / n_probe
        repeats = int(math.ceil(ranks.max(min_time * 1.15 / max(1e-7, per_step * steps))))
        repeats = max(1, min(repeats, 20000))
    n_steps = steps * repeats
    kern_us[:] = 0.0
    n_timed[0] = 0
    # N_REGIONS timed regions of n_steps steps each, every one bracketed by barriers; the line reports the MEDIAN region
    # (value, ms_per_step) and all of them (`regions`), so that a reader sees the run's own spread next to the figure
    regions = []
    for _ in range(N_REGIONS if min_time > 0 else 1):
        ranks.barrier()
        t0 = time.perf_counter()
        run_steps(n_steps, warmup)
        el_local = time.perf_counter() - t0
        ranks.barrier()
        regions.append((ranks.max(time.perf_counter() - t0), el_local))
    regions_sorted = sorted(regions)
    elapsed, elapsed_local = regions_sorted[len(regions_sorted) // 2]
    kern_us /= max(1, n_timed[0])
    # the same kernel with nothing else on the GPU (one batch at a time): with several batches in flight the events of
    # the timed region also span time the kernel shares the chip with the other streams' small kernels, and
    # rocprofv3 serialises dispatches, so this is the figure its kernel stats reproduce
    n_iso = min(steps, 50)
    kern_iso = 0.0
    ctx.set_profiling(1)
    for i in range(n_iso):
        r = packed[(warmup + i) % n_unique]
        ctx.score_packed(r[0], r[1], r[2], r[3], r[4], r[5])
        kern_iso += ctx.last_timing()[0]
    ctx.set_profiling(0)
    kern_iso /= max(1, n_iso)
    # ... and once more with HIP events on the stream the kernels are launched on: bsr_set_profiling(2) times a batch
    # kernel by kernel, which sends it through the slot's HIP stream even where batches are otherwise dispatched as
    # packets (there is no stream for an event to sit on in that path: `kern_us` above is then the dispatch packet's own
    # start / end timestamps).  A cross-check of the clock, a few microseconds above it by the event pair's own cost.
    ev_us = []
    n_ev = min(steps, 24)
    ctx.set_profiling(2)
    for i in range(n_ev):
        r = packed[(warmup + i) % n_unique]
        ctx.score_packed(r[0], r[1], r[2], r[3], r[4], r[5])
        ev_us.append(float(ctx.last_timing()[0]))
    ctx.set_profiling(0)
    # (median, the first four dropped: the HIP runtime loads its own copy of a kernel at that kernel's first launch
    # through it, and none had been made)
    ev_us = sorted(ev_us[4:]) or [0.0]
    kern_ev = ev_us[len(ev_us) // 2]
    verified = verify_timed_results(wl, n_steps, warmup)
    return dict(elapsed=elapsed, elapsed_local=elapsed_local, n_steps=n_steps, repeats=repeats,
                region_elapsed=[r[0] for r in regions],
                kern_us_region=float(kern_us[0]), kern_us=float(kern_iso), kern_us_hip_events=float(kern_ev), verified=verified)


def verify_timed_results(wl, n_steps, first):
    """The line proves its own work: every batch of the timed region wrote its scores into its own result array (the
    last occurrence of each batch is what the arrays hold now); a few of them are scored again, one at a time with
    nothing else in flight, and must come out byte for byte the same.  A pipelined region that skipped work, raced or
    returned stale records fails here, and the bench exits non-zero."""
    import numpy as np
    ctx, packed = wl["ctx"], wl["packed"]
    n_unique = len(packed)
    timed = sorted({(first + i) % n_unique for i in range(max(0, n_steps - n_unique), n_steps)})
    picks = timed[:: max(1, len(timed) // 4)][:4]
    for bi in picks:
        r = packed[bi]
        got = r[5].copy()                       # what the pipelined region returned for this batch
        if not np.isfinite(got["loglik"][got["rank"] == wl["K"]]).all() or (got["rank"] == 0).all():
            raise SystemExit("bench.py: timed batch %d returned no scores" % bi)
        again = np.zeros_like(got)
        ctx.score_packed(r[0], r[1], r[2], r[3], r[4], again)
        if again.tobytes() != got.tobytes():
            bad = int(np.sum(again["loglik"].view(np.uint64) != got["loglik"].view(np.uint64)))
            raise SystemExit("bench.py: batch %d of the timed region differs from its synchronous rescoring "
                             "(%d of %d log-likelihoods)" % (bi, bad, len(got)))
    return {"batches_rescored": len(picks), "byte_identical": True}


FROM_PROFILES_NOTE = ("everything in this object is READ FROM COMMITTED FILES under profiles/ (rocprofv3 kernel stats and "
                      "PMC passes of this same command on an earlier box), not produced by this run; what this run measured "
                      "is outside it")
SHADER_CLOCK_GHZ = 2.4   # under load (tools/tile_stamps.py: 2.41-2.43 GHz with eight batches in flight)


def parity_exemptions():
    """How many proposals / chains of the GPU suite miss the 1e-6 / 1e-7 parity bound and pass through the pinned
    allowlist (tests/golden/exemption_allow.json; every one a tree whose value is chaotic at the ulp level: the oracle's own
    number moves by more than the tolerance under a one-ulp perturbation of X -- tools/verify_exemptions.py)."""
    try:
        allow = json.load(open(os.path.join(ROOT, "tests", "golden", "exemption_allow.json")))
        n = 0
        for k, v in allow.items():
            ids = v.get("ids") if isinstance(v, dict) else v
            n += len(ids) if isinstance(ids, (list, dict)) else 0
        return {"count": n, "source": "tests/golden/exemption_allow.json",
                "bound": "1e-6 relative on the log-likelihood (1e-7 / 1e-5 on config 1's Beta / RMSE histories)",
                "note": "accept decisions and accepted-tree sequences are bit-exact for these too"}
    except Exception as exc:
        return {"error": repr(exc)}


def occupancy_leg(wl, ranks, depth):
    """CU occupancy of the tile row pass by WORKGROUP LIFETIME at this run's pipelined regime, un-traced: a second context
    with the library's stamp ring on (BSR_TILE_STAMPS: every wave's start / end on the 100 MHz clock and the CU it ran on),
    the same chain state and batches, the same number in flight (tools/cu_occupancy.py)."""
    import importlib.util
    from bsr.tape import flatten
    spec = importlib.util.spec_from_file_location("cu_occupancy", os.path.join(ROOT, "tools", "cu_occupancy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    roots = [[flatten(ch.roots[k]) for k in range(wl["K"])] for ch in wl["chains"]]
    # three windows of the ring's last 192 launches (~2 ms each: one scheduling hiccup of the host inside a window moved
    # the figure from 0.70 to 0.52 between two runs of the same build); the median one is reported, all three listed
    runs = [mod.measure(wl["X"], wl["y"], wl["K"], roots, wl["packed"][:32], depth=depth, steps=2500, ring=192,
                        device=ranks.device()) for _ in range(3)]
    ok = [r for r in runs if "error" not in r]
    if not ok:
        return runs[0]
    ok.sort(key=lambda r: r["occupancy_of_256_cus"])
    res = dict(ok[len(ok) // 2])
    res["occupancy_each_window"] = [r["occupancy_of_256_cus"] for r in runs if "error" not in r]
    res["step_us_each_window"] = [r["step_us"] for r in runs if "error" not in r]
    return res


def refuse_debug_knobs():
    """Timing experiments and ablations change what a step does: a benchmark line is not produced under them.
    Returns the BSR_* settings in force (they go into the line)."""
    knobs = {k: v for k, v in os.environ.items() if k.startswith("BSR_")}
    bad = [k for k in knobs if k.startswith("BSR_DEBUG") or k.startswith("BSR_ABLATE")]
    if bad:
        sys.stderr.write("bench.py: refusing to benchmark under %s\n" % ", ".join(sorted(bad)))
        raise SystemExit(2)
    return knobs


def summarize(wl, tr, ranks, args):
    """Throughput and the roofline object of the dominant kernel (tree-eval + projection pass), SURVEY 8d formula:
    bytes = s * N * (|F| + 1 + C_r + C_w): features referenced, y, cached basis columns per chain, C_w = 0."""
    import numpy as np
    N, K, C, P = wl["N"], wl["K"], wl["C"], wl["P"]
    s = 8 if args.dtype == "f64" else 4
    n_feat = float(np.mean(wl["feat_counts"]))
    alg_bytes = int(s * N * (n_feat + 1 + C * (K - 1)))
    p1 = max(tr["kern_us"], 1e-3) * 1e-6
    total_props = ranks.world * P * tr["n_steps"]
    n_unique = len(wl["packed"])
    return {
        "value": total_props / tr["elapsed"],
        "ms_per_step": 1e3 * tr["elapsed"] / tr["n_steps"],
        # `steps` x timed_repeats steps per timed region; N_REGIONS regions, the median one is the line's value
        "timed_repeats": tr["repeats"], "steps_per_timed_region": tr["n_steps"], "timed_region_s": tr["elapsed"],
        "settle_s": SETTLE_S,
        "regions": {"n": len(tr["region_elapsed"]),
                    "value_each": [total_props / e for e in tr["region_elapsed"]],
                    "spread": (max(tr["region_elapsed"]) - min(tr["region_elapsed"])) / tr["elapsed"]},
        "config": {"workload": wl["W"]["desc"], "N": N, "d": wl["d"], "K": K, "chains_per_gpu": C,
                   "proposals_per_step_per_gpu": P, "speculative_batch": wl["B"],
                   "parallelism": "chains x%d" % ranks.world,
                   "avg_nodes_per_tape": wl["n_nodes"] / (n_unique * P),
                   "transcendental_node_frac": wl["n_trans"] / max(1, wl["n_nodes"])},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / p1 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / p1 / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "row pass (tree-eval + projection)", "kernel_us": tr["kern_us"],
                     "kernel_us_in_timed_region": tr["kern_us_region"],
                     "kernel_us_hip_events": tr.get("kern_us_hip_events"),
                     "algorithmic_bytes": alg_bytes, "features_per_launch": n_feat,
                     # the launches of consecutive batches overlap (the tile pass is launched narrower than the machine
                     # for that): the same bytes over the pipelined step, for comparison with `achieved`
                     "achieved_per_pipelined_step": alg_bytes / (tr["elapsed"] / tr["n_steps"]) / 1e9},
    }


def attach_traffic(out, name, B, C, dtype):
    """HBM traffic per launch of the row pass from the committed rocprofv3 PMC run of this same command
    (tools/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH x2 on gfx950)."""
    import glob
    tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_%s_B%d.json" % (name, B))))
    tpath = tpaths[-1] if tpaths else ""
    if C == WORKLOADS[name]["chains"] and dtype == "f64" and tpath:
        recs = [(rec.get("launches", 0), rec) for kname, rec in json.load(open(tpath)).items()
                if "k_rows" in kname or "k_tile" in kname or "k_stream" in kname]
        if recs:   # the scoring pass is the kernel with (by far) the most launches; the others filled derived columns
            rec = max(recs, key=lambda t: t[0])[1]
            # (`traffic` is the contract's field; it cannot be measured in this run -- the counters need passes of their
            # own under rocprofv3 --pmc -- so it is read from the committed profile of this same command, and says so)
            out["roofline"]["traffic"] = rec["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = os.path.relpath(tpath, ROOT) + " (committed rocprofv3 --pmc passes of this command, NOT this run)"
    # the profiler's average duration of the same kernel from the committed kernel-stats run of this command
    # (HIP events around one launch read 2-3 us more: the pair's own cost and the launch gap)
    spaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_bench_%s_B%d.csv" % (name, B))))
    if C == WORKLOADS[name]["chains"] and dtype == "f64" and spaths:
        import csv
        best = None
        for row in csv.DictReader(open(spaths[-1])):
            kn = row.get("Name", "")
            if "k_tile" in kn or "k_stream" in kn or ("k_rows" in kn and ", 0>" in kn):
                if best is None or int(row["Calls"]) > int(best["Calls"]):
                    best = row
        if best:
            fp = out["roofline"].setdefault("from_profiles", {"note": FROM_PROFILES_NOTE})
            fp["kernel_us_rocprofv3"] = float(best["AverageNs"]) / 1e3
            fp["kernel_stats_source"] = os.path.relpath(spaths[-1], ROOT)


def attach_valu(out, name, B, C, dtype, n_cu_used=None):
    """The secondary bound ("state honestly"): instruction issue of the row pass from the committed counter
    run of this same command (tools/pmc_tile.sh).  On gfx950 a SIMD issues one fp64 vector instruction per ~4.4 cycles
    and the CU's ONE scalar unit about one scalar instruction (SALU, branch, scalar load) per cycle for all sixteen
    waves (tools/micro/fp64_issue.hip, issue_mix.hip): both floors are stated, with the share of the kernel's time they
    account for."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_tile_%s_B%d.json" % (name, B))))
    if not (C == WORKLOADS[name]["chains"] and dtype == "f64" and paths):
        return
    recs = json.load(open(paths[-1]))
    if not recs:
        return
    kname, rec = max(recs.items(), key=lambda kv: kv[1].get("launches", 0))
    valu = rec.get("SQ_INSTS_VALU")
    if not valu:
        return
    clock_ghz = 2.3
    waves = rec.get("SQ_WAVES", 0)
    n_cu = n_cu_used or max(1, int(round(waves / 16.0)))
    scalar = rec.get("SQ_INSTS_SALU", 0) + rec.get("SQ_INSTS_BRANCH", 0) + rec.get("SQ_INSTS_SMEM", 0)
    fp64 = rec.get("SQ_INSTS_VALU_ADD_F64", 0) + rec.get("SQ_INSTS_VALU_MUL_F64", 0) + rec.get("SQ_INSTS_VALU_FMA_F64", 0)
    fp = out["roofline"].setdefault("from_profiles", {"note": FROM_PROFILES_NOTE})
    kern_us = fp.get("kernel_us_rocprofv3") or out["roofline"]["kernel_us"]
    valu_us = valu * 4.4 / (4.0 * n_cu) / (clock_ghz * 1e3)
    scalar_us = scalar * 1.2 / n_cu / (clock_ghz * 1e3)
    fp["valu"] = {
        "kernel": kname, "insts": valu, "fp64_insts": fp64, "fp64_share": round(fp64 / valu, 3),
        "scalar_insts": scalar, "cus": n_cu,
        "issue_cycles": int(valu * 4.4 / (4.0 * n_cu)),                 # per SIMD, at 4.4 cycles per fp64 vector instruction
        "valu_floor_us": round(valu_us, 1), "scalar_floor_us": round(scalar_us, 1),
        "busy_frac": round(valu_us / kern_us, 3), "scalar_busy_frac": round(scalar_us / kern_us, 3),
        "lds_bank_conflict_frac": (round(rec["SQ_LDS_BANK_CONFLICT"] / rec["SQ_LDS_IDX_ACTIVE"], 3)
                                   if rec.get("SQ_LDS_IDX_ACTIVE") else None),
        "wait_frac": round(rec["SQ_WAIT_ANY"] / rec["SQ_WAVE_CYCLES"], 3) if rec.get("SQ_WAVE_CYCLES") else None,
        "source": os.path.relpath(paths[-1], ROOT)}
    # ... and flat, next to `frac`: what actually bounds the kernel.  frac is the HBM figure the contract asks for; the
    # kernel is an interpreter, and its time goes to vector issue first.  bound_frac = the largest of the three
    # floors -- HBM time of the PHYSICAL traffic at peak, vector issue, scalar issue -- over the kernel's duration.
    r = fp
    traffic = out["roofline"].get("traffic") or out["roofline"]["algorithmic_bytes"]
    hbm_us = traffic / (HBM_PEAK_GBS * 1e3)
    floors = {"hbm": hbm_us, "valu": valu_us, "scalar": scalar_us}
    top = max(floors, key=floors.get)
    r["valu_insts_per_launch"] = valu
    r["valu_fp64_share"] = round(fp64 / valu, 3)
    r["valu_busy_frac"] = round(valu_us / kern_us, 3)
    r["scalar_busy_frac"] = round(scalar_us / kern_us, 3)
    r["hbm_busy_frac_physical"] = round(hbm_us / kern_us, 3)
    r["bound_frac"] = round(floors[top] / kern_us, 3)
    r["bound_by"] = top
    r["physical_GBps"] = round(traffic / (kern_us * 1e3), 1)
    r["workgroups"] = n_cu
    r["cu_us_per_batch"] = round(kern_us * n_cu, 0)
    # Chip-wide vector issue at the PIPELINED step this run measured: the launch's vector instructions (a property of the
    # workload and the build, counted by the committed PMC run) x 4 cycles / (1 024 SIMDs x shader clock x the step).
    # The true bound of the headline: the tape loop is fp64 vector issue, not HBM.
    step_us = out.get("ms_per_step", 0) * 1e3
    if step_us > 0:
        out["roofline"]["valu_issue_frac_pipelined"] = round(valu * 4.0 / (1024.0 * SHADER_CLOCK_GHZ * 1e3 * step_us), 3)
        out["roofline"]["valu_issue_frac_pipelined_inputs"] = {
            "valu_insts_per_launch": valu, "insts_source": os.path.relpath(paths[-1], ROOT) + " (committed; not this run)",
            "step_us": round(step_us, 3), "step_source": "this run", "simds": 1024, "shader_clock_ghz": SHADER_CLOCK_GHZ,
            "cycles_per_fp64_vector_inst": 4}


def gather_trees(wl, ranks):
    """The one exchange of the path: every chain's current (accepted) trees, all-gathered over RCCL."""
    import numpy as np
    from bsr.dist import pack_chain_record, gather_raw
    if ranks.world <= 1:
        return None
    got = gather_raw(ranks.gather, [pack_chain_record(ch) for ch in wl["chains"]])
    assert len(got) == ranks.world * wl["C"], len(got)
    return len(got)


def engine_leg(args, ranks, chains=8, batch=None, seconds=3.0):
    """End-to-end chain throughput: the native sampler drives `chains` chains on this rank's GPU (consumed MH proposals
    per second, host proposal generation and accept path included), then the accepted trees of all ranks are gathered.
    chains=8: config 4's per-GPU share; chains=1: config 2's single chain; batch=None: the library's default for the data set
    (bsr.native.default_batch: 64 here)."""
    import numpy as np
    from bsr import dist as D
    from bsr.chain import DeviceScorer
    from bsr.native import NativeEngine
    W = WORKLOADS["c4"]
    X, y = synth(W["N"], W["d"], seed=0)
    from bsr.native import batch_shape, default_batch
    if not batch:      # what BSR.fit / bsr.sharded.run_rank take where the caller names none
        batch = default_batch(W["N"], W["d"], W["K"])
    tc, tb = batch_shape(chains, batch, W["K"])     # (as BSR.fit and bsr.sharded create their contexts)
    scorer = DeviceScorer(X, y, W["K"], n_chains=chains, max_batch=chains * batch, device=ranks.device(),
                          dtype=args.dtype, typical_chains=tc, typical_batch=tb)
    eng = NativeEngine(scorer.ctx, chains, W["d"], val=10 ** 9)
    eng.set_nan_policy(True)     # a throughput leg: a NaN candidate is a rejection, not the reference's LinAlgError
    try:
        for c in range(chains):
            eng.seed(c, 1000 + ranks.rank * chains + c)
            eng.init_chain(c)
        eng.run(batch_per_chain=batch, max_props=200)            # warm-up
        # ONE call of `seconds` (as BSR.fit makes one): a calibration run sizes it.  (Until round 6 the leg called run()
        # in steps of 4 000 proposals per chain -- 8 ms each, a fifth of which went into starting and draining the worker
        # threads' pipelines 375 times.)
        t0 = time.perf_counter()
        eng.run(batch_per_chain=batch, max_props=4200)
        rate = 4000.0 / max(1e-6, time.perf_counter() - t0)       # proposals per chain and second
        done0 = sum(eng.result(c)["n_props"] for c in range(chains))
        ranks.barrier()
        t0 = time.perf_counter()
        eng.run(batch_per_chain=batch, max_props=4200 + max(4000, int(rate * seconds)))
        ranks.barrier()
        dt = ranks.max(time.perf_counter() - t0)
        res = [eng.result(c, current=True) for c in range(chains)]
        memo = [eng.memo_stats(c) for c in range(chains)]
        done = sum(r["n_props"] for r in res) - done0
        recs = [D.pack_record(ranks.rank * chains + c, None, r["beta"], r["sigma"], r["errs"], r["n_props"],
                              r["n_accept"], r["n_rank_rejects"], r["n_discarded"], tapes_in=r["tapes"])
                for c, r in enumerate(res)]
        n_gathered = None
        total = done
        if ranks.world > 1:
            n_gathered = len(D.gather_raw(ranks.gather, recs))
            total = float(np.sum(ranks.gather.allgather(np.array([done], dtype=np.float64).view(np.uint8))
                                 .reshape(-1).view(np.float64)))
        return {"metric": "consumed MH proposals/s, native sampler, %d chains x batch %d per GPU" % (chains, batch),
                "value": total / dt, "seconds": dt, "chains_total": chains * ranks.world,
                "gathered_records": n_gathered,
                # CONSUMED proposals: a chain's repeats of a candidate already scored in its current state are answered
                # from the sampler's memo (they are part of what a chain consumes, not of what the GPU scored: the
                # headline counts GPU-scored proposals only)
                "memo_answered_fraction_of_generated": sum(m[0] for m in memo) / max(1, sum(m[1] for m in memo)),
                "discarded_fraction": sum(r["n_discarded"] for r in res) /
                max(1, sum(r["n_discarded"] + r["n_props"] for r in res))}
    finally:
        eng.close()
        scorer.close()


def deep_leg(args, ranks, n_batches=12):
    """BASELINE configs[4] says "deep trees (depth <= 12)"; the real move mix of a burnt-in chain averages 3-4 nodes per
    tape, so the c5 leg never shows how the streaming pass behaves on them.  This leg scores batches of 64 candidates
    GROWN to height 8..12 (the reference's own grow(), codes/funcs.py:74-119, with its depth prior flattened: beta -0.15
    instead of -1, at most 400 nodes) on the c5 data set and reports nodes per tape, how many tapes the assembly
    interpreter takes (round 6: programs of several words, a second value below the accumulator) and
    the row pass's duration per launch."""
    import numpy as np
    from bsr import grow
    from bsr.node import Node, getHeight, getNum
    from bsr.proposal import OpTable
    from bsr.tape import flatten, pack
    short = argparse.Namespace(**vars(args))
    short.batch, short.chains = 0, 0
    wl = build_workload("c5", short, ranks)
    generate_batches(wl, 2)
    ctx, B, d, K = wl["ctx"], wl["B"], wl["d"], wl["K"]
    ops = ['inv', 'ln', 'neg', 'sin', 'cos', 'exp', 'square', 'cubic', '+', '*']
    w, typ = [0.1] * 10, [1] * 8 + [2, 2]
    np.random.seed(4242)
    trees = []
    while len(trees) < B * n_batches:
        root = Node(0)
        grow(root, d, ops, w, typ, -0.15, 1.0, 1.0)
        h, n = getHeight(root), getNum(root)
        if 8 <= h <= 12 and n <= 400:
            trees.append((root, h, n))
    base = wl["packed"][0]
    heights = [t[1] for t in trees]
    nodes = [t[2] for t in trees]
    packed = []
    for b in range(n_batches):
        rows, off = pack([flatten(t[0]) for t in trees[b * B:(b + 1) * B]])
        out = np.zeros(B, dtype=wl["lib"].SCORE_DTYPE)
        packed.append((rows, off, base[2], base[3], base[4], out))
    ctx.set_profiling(1)
    kern, stats, n_nan = [], {"tapes": 0, "asm_program_tapes": 0, "chain_tapes": 0, "stream_entries": 0}, 0
    for rep in range(3):
        for r in packed:
            t = ctx.score_submit(r[0], r[1], r[2], r[3], r[4])
            ctx.score_wait(t, r[5])
            if rep > 0:
                kern.append(float(ctx.last_timing()[0]))
            if rep == 2:
                st = ctx.batch_stats(t)
                for k in stats:
                    stats[k] += st[k]
                n_nan += int(np.sum(r[5]["rank"] < 0))
    ctx.set_profiling(0)
    # pipelined throughput over the same batches
    t0 = time.perf_counter()
    tickets, n_steps = [], 0
    while time.perf_counter() - t0 < 0.6:
        r = packed[n_steps % n_batches]
        tickets.append((ctx.score_submit(r[0], r[1], r[2], r[3], r[4]), r))
        if len(tickets) >= 4:
            t, rr = tickets.pop(0)
            ctx.score_wait(t, rr[5])
        n_steps += 1
    while tickets:
        t, rr = tickets.pop(0)
        ctx.score_wait(t, rr[5])
    dt = time.perf_counter() - t0
    info = ctx.info()
    wl["scorer"].close()
    kern_us = float(np.median(kern))
    s = 8
    alg_bytes = int(s * wl["N"] * (min(d, 50) + 1 + (K - 1)))          # every feature is referenced by tapes this long
    return {"metric": "MH proposals scored/sec, candidates grown to height 8..12 (N=1M, d=50, K=3)",
            "value": n_steps * B / dt, "unit": "proposals/s", "ms_per_step": 1e3 * dt / n_steps,
            "avg_nodes_per_tape": float(np.mean(nodes)), "max_nodes_per_tape": int(np.max(nodes)),
            "height_min_max": [int(np.min(heights)), int(np.max(heights))],
            "avg_stream_entries_per_tape": stats["stream_entries"] / max(1, stats["tapes"]),
            "share_of_tapes_for_the_assembly_interpreter": stats["asm_program_tapes"] / max(1, stats["tapes"]),
            "share_of_chain_tapes": stats["chain_tapes"] / max(1, stats["tapes"]),
            "nan_candidates": n_nan,
            "row_pass": info["row_pass"], "row_pass_us_per_launch": kern_us,
            "roofline": {"bound": "hbm", "achieved": alg_bytes / (kern_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes,
                         "note": "an interpreter at ~%d entries per tape is bound by instruction issue, not by HBM" %
                                 round(stats["stream_entries"] / max(1, stats["tapes"]))}}


def f32_leg(args, ranks):
    """BASELINE configs[4]: the fp32 context (f32 storage; where the slices stream -- this config -- f64 arithmetic on the stored values) against the fp64 one at
    N = 1M, d = 50: throughput and roofline of the f32 row pass (s = 4 bytes), and -- on the SAME chain state and the
    SAME 64 real-mix proposals -- how far its log-likelihoods, rank decisions and accept decisions move."""
    import numpy as np
    from bsr.device import DeviceContext
    from bsr.tape import flatten
    short = argparse.Namespace(**vars(args))
    short.batch, short.chains = 0, 0
    short.dtype = "f32"
    wl = build_workload("c5", short, ranks)
    generate_batches(wl, 24)
    steps = max(10, min(args.steps, 40))
    tr = timed_region(wl, ranks, steps, 5, args.depth, min(args.min_time, 0.5))
    res = summarize(wl, tr, ranks, short)
    res["unit"], res["steps"], res["verified"] = "proposals/s", steps, tr["verified"]
    # the comparison: this (f32-driven) chain's state and one of its batches through an f64 context as well
    r = wl["packed"][0]
    b32 = np.zeros_like(r[5])
    wl["ctx"].score_packed(r[0], r[1], r[2], r[3], r[4], b32)
    ch = wl["chains"][0]
    ctx64 = DeviceContext(wl["X"], wl["y"], K=wl["K"], n_chains=1, max_batch=wl["B"], device=ranks.device(), dtype="f64")
    for k in range(wl["K"]):
        ctx64.set_current(0, k, flatten(ch.roots[k]))
    ctx64.refresh(0)
    a64 = np.zeros_like(r[5])
    ctx64.score_packed(r[0], r[1], r[2], r[3], r[4], a64)
    ctx64.close()
    wl["scorer"].close()
    K = wl["K"]
    both = (a64["rank"] == K) & (b32["rank"] == K)
    d_abs = np.abs(b32["loglik"][both] - a64["loglik"][both])
    rel = d_abs / np.abs(a64["loglik"][both])
    # the accept test compares log u with log R, in which only the likelihood term depends on the dtype
    # (codes/funcs.py:1249-1254): a decision can flip only where |delta loglik| reaches the margin |log R - log u|, which
    # is O(1) for almost every proposal -- count the proposals whose log-likelihood moves by more than 1e-2
    res["vs_f64"] = {"proposals": int(len(a64)), "both_full_rank": int(both.sum()),
                     "rel_dloglik_median": float(np.median(rel)) if rel.size else None,
                     "rel_dloglik_max": float(np.max(rel)) if rel.size else None,
                     "abs_dloglik_max": float(np.max(d_abs)) if rel.size else None,
                     "rank_flips": int(np.sum(a64["rank"] != b32["rank"])),
                     "rank_flips_towards_deficient": int(np.sum((a64["rank"] == K) & (b32["rank"] < K))),
                     "accept_flip_candidates_abs_dloglik_over_1e-2": int(np.sum(d_abs > 1e-2))}
    return res


def unpinned_leg(args):
    """The headline workload once more WITHOUT BSR_PIN=1 (this script confines itself to the library's CPUs; a drop-in
    caller does not): a child process, after this one's contexts are closed."""
    import subprocess
    env = dict(os.environ)
    env["BSR_PIN"] = "0"
    cmd = [sys.executable, os.path.abspath(__file__), "--extras", "0", "--cpu-sample", "0", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--min-time", str(min(args.min_time, 0.5))]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "env_knobs": d.get("env_knobs"),
            "caller_cpus": d["per_rank"][0]["caller_cpus"], "verified": d.get("verified")}


def main():
    args = parse_args()
    knobs = refuse_debug_knobs()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_children(args)
    ranks = Ranks()
    if args.gpus != ranks.world:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); start it as `python bench.py --gpus N` "
                         "or under a launcher with --nproc-per-node N\n" % (args.gpus, ranks.world))
        return 2
    extras = args.extras if args.extras >= 0 else int(args.workload == "c2" and not args.batch and not args.chains
                                                     and args.dtype == "f64")
    cpu_mp = None
    if ranks.rank == 0 and args.cpu_sample > 0 and (args.workload == "c4" or (args.chains or 0) > 1):
        cpu_mp = cpu_baseline_multiproc(args.cpu_sample / 2)      # before this process creates a GPU context

    ranks.connect()
    wl = build_workload(args.workload, args, ranks)
    n_unique = min(args.warmup + args.steps, max(64, args.warmup + 44))
    generate_batches(wl, n_unique)
    tr = timed_region(wl, ranks, args.steps, args.warmup, args.depth, args.min_time)
    out = {"metric": "MH proposals scored/sec (N=100k,d=10,K=3) at 1/2/4/8 MI355X; HBM GB/s",
           "value": None, "unit": "proposals/s", "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic"}
    out.update(summarize(wl, tr, ranks, args))
    out["batches_in_flight"] = max(1, min(8, pick_depth(args.depth, wl["K"])))
    # (also inside `config`: the driver's record keeps the contract's objects and only the NAMES of the other keys)
    out["config"].update(timed_region_s=tr["elapsed"], timed_repeats=tr["repeats"], steps_per_timed_region=tr["n_steps"],
                         batches_in_flight=out["batches_in_flight"])
    out["parity_exemptions"] = parity_exemptions()
    out["config"]["parity_exemptions"] = out["parity_exemptions"].get("count")
    out["verified"] = tr["verified"]
    # per rank: its own step time, the threads and CPUs the library chose for it (what an 8-rank run on a 16-CPU quota
    # actually gets: DESIGN 6)
    info = wl["ctx"].info()
    mine = {"rank": ranks.rank, "ms_per_step": 1e3 * tr["elapsed_local"] / tr["n_steps"], "submit_threads": info["submit_threads"],
            "lib_cpus": info["lib_cpus"], "cpu_budget": info["cpu_budget"], "caller_cpus": len(os.sched_getaffinity(0)),
            "gpu_numa_node": info["gpu_numa_node"]}
    if ranks.world > 1:
        import numpy as np
        vec = np.array([mine["rank"], mine["ms_per_step"], mine["submit_threads"], mine["lib_cpus"], mine["cpu_budget"],
                        mine["caller_cpus"]], dtype=np.float64)
        got = ranks.gather.allgather(vec.view(np.uint8)).reshape(ranks.world, -1).view(np.float64)
        out["per_rank"] = [{"rank": int(g[0]), "ms_per_step": float(g[1]), "submit_threads": int(g[2]), "lib_cpus": int(g[3]),
                            "cpu_budget": float(g[4]), "caller_cpus": int(g[5])} for g in got]
    else:
        out["per_rank"] = [mine]
    out["env_knobs"] = {k: v for k, v in sorted(knobs.items()) if k not in ("BSR_SHARE_DEVICE",)}
    attach_traffic(out, args.workload, wl["B"], wl["C"], args.dtype)
    attach_valu(out, args.workload, wl["B"], wl["C"], args.dtype)
    # which row pass and geometry the context chose (tests/test_gpu_regimes.py pins it per BASELINE config)
    out["config"]["geometry"] = {k: info[k] for k in ("row_pass", "tape_groups", "row_slices", "blocks_per_slice")}
    # how the batches reached the GPU: AQL packets into the library's own queues (csrc/bsr_aql.h) or HIP launches on the
    # slots' streams -- and with it the clock of `kernel_us`: the packet processor's start / end timestamps of the row
    # pass's dispatch packet (read through hsa_amd_profiling_get_dispatch_time, what rocprofv3 reads too), or HIP events
    # on the stream the kernel was launched on
    disp = wl["ctx"].dispatch_info()
    out["dispatch"] = disp
    out["roofline"]["kernel_us_clock"] = ("dispatch packet timestamps (direct AQL dispatch: no HIP stream on the path)"
                                          if disp["batches_direct"] > disp["batches_streamed"] else "HIP events on the launch stream")
    if ranks.rank == 0 and extras and info["row_pass"] == "k_tile1a":
        try:
            occ = occupancy_leg(wl, ranks, out["batches_in_flight"])
            out["roofline"]["cu_occupancy_by_wg_lifetime"] = occ.get("occupancy_of_256_cus")
            out["roofline"]["cu_occupancy"] = occ
        except Exception as exc:
            out["roofline"]["cu_occupancy"] = {"error": repr(exc)}
    n_g = gather_trees(wl, ranks)
    if n_g is not None:
        out["gathered_records"] = n_g
        out["gather"] = ranks.via
    if ranks.rank == 0 and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(wl["X"], wl["y"], wl["K"], wl["chains"], wl["packed"][args.warmup:],
                                           args.cpu_sample)
        if cpu_mp is not None:
            out["cpu_baseline"]["multiprocess"] = cpu_mp
    wl["scorer"].close()

    if extras:
        ex = {}
        short = argparse.Namespace(**vars(args))
        short.batch, short.chains = 0, 0
        legs = ["c3", "c5", "c4"] if ranks.world == 1 else ["c4"]
        for name in legs:
            try:
                w2 = build_workload(name, short, ranks)
                steps2 = max(10, min(args.steps, 60))
                generate_batches(w2, min(args.warmup + steps2, 40))
                t2 = timed_region(w2, ranks, steps2, min(args.warmup, 10), args.depth, min(args.min_time, 0.5))
                s2 = summarize(w2, t2, ranks, args)
                s2["unit"] = "proposals/s"
                s2["verified"] = t2["verified"]
                s2["steps"] = steps2
                attach_traffic(s2, name, w2["B"], w2["C"], args.dtype)
                attach_valu(s2, name, w2["B"], w2["C"], args.dtype)
                i2 = w2["ctx"].info()
                s2["config"]["geometry"] = {k: i2[k] for k in ("row_pass", "tape_groups", "row_slices", "blocks_per_slice")}
                if ranks.world > 1:
                    s2["gathered_records"] = gather_trees(w2, ranks)
                ex[name] = s2
                w2["scorer"].close()
            except Exception as exc:                               # an extra leg never costs the headline line
                ex[name] = {"error": repr(exc)}
        try:
            ex["c4_native_engine"] = engine_leg(args, ranks)
        except Exception as exc:
            ex["c4_native_engine"] = {"error": repr(exc)}
        try:
            ex["c2_native_engine"] = engine_leg(args, ranks, chains=1, seconds=2.0)   # (BSR.fit's default batch)
            # what a single chain CONSUMES next to what the headline scores (speculative batches of a frozen state):
            out["consumed_per_s"] = ex["c2_native_engine"]["value"]
        except Exception as exc:
            ex["c2_native_engine"] = {"error": repr(exc)}
        if ranks.world == 1:
            try:
                ex["c5_deep"] = deep_leg(args, ranks)
            except Exception as exc:
                ex["c5_deep"] = {"error": repr(exc)}
            try:
                ex["c5_f32"] = f32_leg(args, ranks)
                import glob
                sw = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fp32_chain_sweep.json")))
                if sw:   # the chain-level sweep (tools/fp32_chain_sweep.py: decisions per consumed proposal, N = 1e4 .. 1e6)
                    ex["c5_f32"]["from_profiles"] = {"note": FROM_PROFILES_NOTE, "chain_sweep": json.load(open(sw[-1]))["rows"],
                                                     "chain_sweep_source": os.path.relpath(sw[-1], ROOT)}
            except Exception as exc:
                ex["c5_f32"] = {"error": repr(exc)}
            try:
                ex["c2_unpinned"] = unpinned_leg(args)
            except Exception as exc:
                ex["c2_unpinned"] = {"error": repr(exc)}
            try:
                # The headline's step is a batch of 64 speculative proposals of ONE chain: about what a chain can use (its
                # sampler keeps 3 x 32 in flight and throws ~8 % away).  The same workload with deeper batches, for the
                # record: a launch's fixed costs -- the row slice staged into LDS, three dependent kernels in one queue --
                # are shared by more tapes, but a lone chain would discard most of a batch this deep.
                sweep = {}
                for b in (128, 256):
                    a2 = argparse.Namespace(**vars(args))
                    a2.batch, a2.chains = b, 0
                    w3 = build_workload("c2", a2, ranks)
                    generate_batches(w3, 24)
                    t3 = timed_region(w3, ranks, 40, 10, args.depth, 0.4)
                    sweep["B%d" % b] = {"value": ranks.world * w3["P"] * t3["n_steps"] / t3["elapsed"],
                                        "us_per_step": 1e6 * t3["elapsed"] / t3["n_steps"], "verified": t3["verified"]}
                    w3["scorer"].close()
                ex["c2_batch_sweep"] = {"metric": "MH proposals scored/sec, one chain, deeper speculative batches (N=100k, d=10, K=3)",
                                        "unit": "proposals/s", "headline_B": out["config"]["speculative_batch"], **sweep}
            except Exception as exc:
                ex["c2_batch_sweep"] = {"error": repr(exc)}
        out["extra"] = ex
    if ranks.rank == 0:
        print(json.dumps(out))
    return 0


def _onode(O, n):
    m = O.ONode(n.depth)
    m.type, m.operator, m.op_ind, m.feature, m.a, m.b = n.type, n.operator, n.op_ind, n.feature, n.a, n.b
    m.left = _onode(O, n.left) if n.left is not None else None
    m.right = _onode(O, n.right) if n.right is not None else None
    return m


def _cpu_faithful(O, Xdf, ys, y, K, chains, batches, budget_s):
    """Reference-faithful flavour: per proposal K+1 tree evaluations with per-element exp/inv loops, SVD rank gate,
    two ylogLike passes (codes/funcs.py:1212-1235)."""
    import numpy as np
    from bsr.tape import unflatten
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
                        new_o[:, j] = O.allcal(_onode(O, unflatten(tapes[i])), Xdf, faithful=True)[:, 0]
                        old_o[:, j] = O.allcal(_onode(O, ch.roots[j]), Xdf, faithful=True)[:, 0]
                    else:
                        col = O.allcal(_onode(O, ch.roots[j]), Xdf, faithful=True)[:, 0]
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
                return done, time.perf_counter() - t0
    return done, time.perf_counter() - t0


def cpu_baseline(X, y, K, chains, batches, budget_s):
    """Times the oracle's restatement of the same scoring work on the host CPU.  Bounded sample: proposals of the
    timed batches, in order.  Three figures: reference-faithful with the BLAS pool at its default size (the headline
    `value`; the reference is one CPython thread whose numpy calls may thread), the same with the BLAS pool limited
    to one thread, and the vectorised-fair flavour."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import pandas as pd
    import bsr_oracle as O
    from bsr.tape import unflatten
    Xdf = pd.DataFrame(X)
    ys = pd.Series(y)
    cores = os.cpu_count() or 1
    done, dt = _cpu_faithful(O, Xdf, ys, y, K, chains, batches, budget_s * 0.45)
    out = {"value": done / dt, "unit": "proposals/s", "cores": 1, "kind": "port",
           "blas_threads": "default (%d logical cores visible)" % cores,
           "sample": "%d proposals of the timed batches in order, oracle reference-faithful flavour "
                     "(K+1 allcal with per-element exp/inv loops + matrix_rank + 2 ylogLike), %.1f s" % (done, dt)}
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            d1, t1 = _cpu_faithful(O, Xdf, ys, y, K, chains, batches, budget_s * 0.3)
        out["blas_1_thread"] = {"value": d1 / t1, "unit": "proposals/s",
                                "sample": "%d proposals, same flavour, BLAS/OpenMP pools limited to 1 thread, %.1f s" % (d1, t1)}
    except Exception as exc:
        out["blas_1_thread"] = {"error": repr(exc)}
    # vectorised-fair flavour: the same CPU path written the way a numpy user would -- vectorised exp/inv,
    # the sibling columns and the old log-likelihood cached per chain -- so the ratio is not only "a Python loop removed"
    t1 = time.perf_counter()
    fair = 0
    cache = {}
    budget2 = budget_s * 0.25
    for batch in batches:
        tapes, chs, ks, sig = batch[6], batch[2], batch[3], batch[4]
        for i in range(len(tapes)):
            ch = chains[int(chs[i])]
            k = int(ks[i])
            with np.errstate(all="ignore"):
                if ch.index not in cache:
                    cols = np.stack([O.allcal(_onode(O, ch.roots[j]), Xdf, faithful=False)[:, 0] for j in range(K)], axis=1)
                    cache[ch.index] = (cols, O.yloglike(ys, cols, ch.sigma) if np.all(np.isfinite(cols)) else None)
                new_o = cache[ch.index][0].copy()
                new_o[:, k] = O.allcal(_onode(O, unflatten(tapes[i])), Xdf, faithful=False)[:, 0]
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


def _mp_worker(args):
    """One CPU process = one chain of the oracle's reference-faithful newProp loop (host driver + scoring)."""
    seed, budget_s = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np
    import bsr_oracle as O
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    W = WORKLOADS["c4"]
    X, y = synth(W["N"], W["d"], seed=0)
    np.random.seed(seed)
    t0 = time.perf_counter()
    n = [0]
    t_last = [t0]

    class Stop(Exception):
        pass

    def on_prop(tr):
        n[0] += 1
        t_last[0] = time.perf_counter()
        if t_last[0] - t0 > budget_s:
            raise Stop()
    try:
        with np.errstate(all="ignore"):
            O.run_chain(X, y, K=W["K"], val=10 ** 9, faithful=True, on_proposal=on_prop)
    except Stop:
        pass
    except Exception:        # e.g. LinAlgError on a NaN candidate, as in the reference: the chain ends there
        pass
    return n[0], t_last[0] - t0


def cpu_baseline_multiproc(budget_s):
    """independent chains are the only CPU parallelism the reference offers -- min(64, cores) processes,
    one chain each, the oracle's faithful newProp loop at config 4's sizes."""
    import multiprocessing as mp
    procs = max(1, min(64, os.cpu_count() or 1))
    try:
        with mp.get_context("spawn").Pool(procs) as pool:
            res = pool.map(_mp_worker, [(1000 + c, budget_s) for c in range(procs)])
        rate = sum(n / max(t, 1e-9) for n, t in res if n > 0)
        return {"value": rate, "unit": "proposals/s", "cores": procs,
                "sample": "%d processes x 1 chain, oracle faithful newProp loop (N=100k, d=10, K=3), ~%.1f s each; "
                          "%d proposals in all" % (procs, budget_s, sum(n for n, _ in res))}
    except Exception as exc:
        return {"error": repr(exc)}


if __name__ == "__main__":
    sys.exit(main())
