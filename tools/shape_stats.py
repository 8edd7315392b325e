#!/usr/bin/env python3
"""Stream-shape statistics of the real move mix (CPU only; test/tool infrastructure, uses the oracle as the scorer).

Reproduces bench.py's batch generation (burn-in, frozen chain state, B speculative proposals per batch) with a CPU
stand-in for the data side at a reduced N, encodes every tape the way csrc/bsr_stage.hip: stage_tapes does
(`terminal f, unary op` -> derived column, `terminal, +|*` -> fused entry) and prints how often each stream shape
occurs -- the input of tools/gen_shapes.py (the straight-line evaluators of csrc/bsr_shapes.h).

    python tools/shape_stats.py [--K 3] [--d 10] [--N 4000] [--batches 64] [--seeds 4] [--json out.json]
"""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-symreg_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import numpy as np

OPN = {0: "inv", 1: "ln", 2: "neg", 3: "sin", 4: "cos", 5: "exp", 6: "sq", 7: "cub", 8: "+", 9: "*", 10: "T",
       11: "+T", 12: "*T", 13: "-", 14: "/", 15: "log"}
DERIVED = (0, 2, 3, 4, 5, 6, 7, 15)


def stream_of(tape, derived=True):
    """opcode stream of one tape after the encoder's two fusions; derived terminals are written 'D'."""
    ops = [int(o) for o in tape["opcode"]]
    out = []
    j, n = 0, len(ops)
    while j < n:
        o = ops[j]
        if o == 10:
            name = "T"
            if derived and j + 1 < n and ops[j + 1] in DERIVED:
                j += 1
            nxt = ops[j + 1] if j + 1 < n else -1
            if out and nxt in (8, 9):
                name = "+T" if nxt == 8 else "*T"
                j += 1
            out.append(name)
        else:
            out.append(OPN[o])
        j += 1
    return tuple(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--N", type=int, default=4000)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--batches", type=int, default=64)
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--burnin", type=int, default=300)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    from bench import synth
    from bsr.chain import Chain, run_chains
    from test_host_driver import OracleScorer
    shapes = collections.Counter()
    lens = collections.Counter()
    n_t = 0
    for sd in range(args.seeds):
        X, y = synth(args.N, args.d, seed=0)
        sc = OracleScorer(X, y, args.K, n_chains=1, max_batch=args.B)
        np.random.seed(1000 + sd)
        ch = Chain(0, sc, args.N, args.d, args.K, val=10 ** 9)
        run_chains([ch], sc, batch_per_chain=args.B, max_props=args.burnin)
        for _ in range(args.batches):
            for cd in ch.generate(args.B):
                s = stream_of(cd.tape)
                shapes[s] += 1
                lens[len(s)] += 1
                n_t += 1
            ch.rng_state = ch._end_state
    print("tapes %d, distinct shapes %d, mean stream length %.2f" % (
        n_t, len(shapes), sum(k * v for k, v in lens.items()) / n_t))
    cum = 0
    for i, (s, c) in enumerate(shapes.most_common(80)):
        cum += c
        print("%3d %6.2f%% cum %6.2f%%  %s" % (i, 100.0 * c / n_t, 100.0 * cum / n_t, " ".join(s)))
    print("length histogram:", sorted(lens.items()))
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"n": n_t, "shapes": [[list(s), c] for s, c in shapes.most_common()]}, f)


if __name__ == "__main__":
    main()
