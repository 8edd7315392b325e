#!/usr/bin/env python3
"""Static ISA audit of the row-pass kernels (CPU only: works on the gfx950 code object build.sh leaves in
csrc/build/bsr_tile.o).

    python tools/isa_audit.py [--kernel k_tile1IdLi3] [--top 12] [--json out.json]

Per function of the code object (the kernel and the out-of-line device functions it calls -- sin/cos/exp/log rows, the
chain evaluator's helpers): instruction counts by class -- fp64 arithmetic (fma / mul / add / other), other vector ALU
(moves, integer, conversions, DPP / permlane lane traffic), LDS (reads / writes / swizzles), vector memory, scalar ALU,
scalar memory, waits, branches -- plus the registers and scratch the compiler reports.  Static counts, not executed
ones: they say what a pass over the code costs (the hot loops are short and straight), and where the non-arithmetic
share sits; the executed totals per launch come from the counters (tools/pmc_tile.sh)."""
import argparse
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "mcmc-symreg_amd", "csrc", "build", "bsr_tile.o")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

CLASSES = [
    ("f64_fma", re.compile(r"^v_fma_f64|^v_fmac_f64")),
    ("f64_mul", re.compile(r"^v_mul_f64")),
    ("f64_add", re.compile(r"^v_add_f64")),
    ("f64_other", re.compile(r"^v_(max|min|rcp|rsq|sqrt|div_|trig|frexp|ldexp|fract|floor|ceil|rndne|trunc|cmp\w*_f64|cmpx\w*_f64)\w*f64|^v_\w+_f64")),
    ("lane_traffic", re.compile(r"^v_permlane|^v_readlane|^v_readfirstlane|^v_writelane|_dpp$|^v_mov_b32_dpp|^v_mov_b64_dpp")),
    ("valu_other", re.compile(r"^v_")),
    ("lds_read", re.compile(r"^ds_read|^ds_load")),
    ("lds_write", re.compile(r"^ds_write|^ds_store")),
    ("lds_swizzle", re.compile(r"^ds_swizzle|^ds_bpermute|^ds_permute")),
    ("lds_other", re.compile(r"^ds_")),
    ("vmem", re.compile(r"^global_|^buffer_|^flat_|^scratch_")),
    ("smem", re.compile(r"^s_load|^s_buffer_load|^s_memtime|^s_memrealtime|^s_dcache")),
    ("wait", re.compile(r"^s_waitcnt|^s_nop|^s_sleep|^s_barrier")),
    ("branch", re.compile(r"^s_cbranch|^s_branch|^s_setpc|^s_swappc|^s_getpc|^s_call|^s_endpgm")),
    ("salu", re.compile(r"^s_")),
]


def classify(mn, ops):
    if ops.endswith("_dpp") or " row_" in ops or "quad_perm" in ops:
        return "lane_traffic"
    for name, rx in CLASSES:
        if rx.search(mn):
            return name
    return "other"


def disassemble():
    if not os.path.exists(OBJ):
        sys.exit("no %s: run mcmc-symreg_amd/csrc/build.sh first" % OBJ)
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(OBJ, os.path.join(tmp, "t.o"))
        subprocess.run([OBJDUMP, "--offloading", "t.o"], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not co:
            sys.exit("no gfx950 code object inside bsr_tile.o")
        return subprocess.run([OBJDUMP, "-d", co[0]], cwd=tmp, check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_tile1IdLi3", help="substring of the kernel's mangled name")
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    funcs = collections.OrderedDict()
    calls = collections.defaultdict(set)
    cur = None
    for line in disassemble().splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            funcs[cur] = collections.Counter()
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//", line)
        if not m or cur is None:
            continue
        mn, ops = m.group(1), m.group(2)
        funcs[cur][classify(mn, mn + " " + ops)] += 1
        funcs[cur]["total"] += 1
    kern = [f for f in funcs if a.kernel in f and "k_tile" in f]
    if not kern:
        sys.exit("no kernel matching %r; kernels: %s" % (a.kernel, [f for f in funcs if "k_tile" in f][:8]))
    order = ["total", "f64_fma", "f64_mul", "f64_add", "f64_other", "valu_other", "lane_traffic", "lds_read", "lds_write",
             "lds_swizzle", "lds_other", "vmem", "salu", "smem", "wait", "branch", "other"]
    print("%-58s" % "function" + "".join("%9s" % c[:9] for c in order))
    out = {}
    def row(name, c):
        print("%-58s" % name[-58:] + "".join("%9d" % c.get(k, 0) for k in order))
        out[name] = {k: c.get(k, 0) for k in order}
    for k in kern:
        row(k, funcs[k])
    # the out-of-line device functions (anything that is not a kernel), largest first
    dev = [(f, c) for f, c in funcs.items() if "k_tile" not in f and c["total"] > 0]
    dev.sort(key=lambda fc: -fc[1]["total"])
    for f, c in dev[:a.top]:
        row(f, c)
    for k in kern:
        c = funcs[k]
        v = sum(c[x] for x in ("f64_fma", "f64_mul", "f64_add", "f64_other", "valu_other", "lane_traffic"))
        f64 = sum(c[x] for x in ("f64_fma", "f64_mul", "f64_add", "f64_other"))
        print("%s: vector ALU %d of %d instructions, fp64 arithmetic %d (%.0f %% of the vector ALU), lane traffic %d, "
              "scalar ALU %d, waits/nops %d" % (k[-40:], v, c["total"], f64, 100.0 * f64 / max(1, v), c["lane_traffic"],
                                                c["salu"], c["wait"]))
    res = os.path.join(os.path.dirname(OBJ), "bsr_tile.resources.txt")
    if os.path.exists(res):
        txt = open(res).read()
        for k in kern:
            m = re.search(re.escape(k) + r".*?VGPRs: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", txt, re.S)
            if m:
                print("%s: VGPRs %s, SGPR spills %s, VGPR spills %s" % (k[-40:], m.group(1), m.group(2), m.group(3)))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
