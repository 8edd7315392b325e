#!/bin/bash
# Interleaved A/B (/C/...) of environment knobs or bench arguments in ONE box (boxes differ by +-4 %; a knob often by less).
# Replaces the one-off tools/probes/r03_*.sh and r05_*.sh scripts of rounds 3-5 (each was this loop with other labels).
#
#   bash tools/ab.sh [-r ROUNDS] [-e] [-o OUT.txt] "label: ENV=1 ENV2=x -- --workload c3 --depth 6" "other: BSR_AQL=0" ...
#
#   -r N   rounds (default 3): every variant once per round, in the order given
#   -e     the native sampler's legs instead of the scoring bench: consumed proposals/s for one chain and eight
#          (bench.engine_leg), memo share and discarded share
#   -o F   also append the lines to F (e.g. gpurun_out/r06/x_ab.txt; copy what is kept into profiles/)
# A variant is "label: [ENV=value ...] [-- bench.py arguments]"; without "--" the default arguments are
# --cpu-sample 0 --extras 0.  Example (round 5's direct-dispatch A/B):
#   bash tools/ab.sh "aql: BSR_AQL=1" "streams: BSR_AQL=0" "aql rows2048: BSR_AQL=1 -- --rows 2048 --min-time 0.5"
rounds=3; engine=0; out=/dev/null
while getopts "r:eo:" o; do case $o in r) rounds=$OPTARG;; e) engine=1;; o) out=$OPTARG;; esac; done
shift $((OPTIND - 1))
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    label="${v%%:*}"; rest="${v#*:}"
    envs="${rest%%--*}"; args=""
    case "$rest" in *--*) args="${rest#*--}";; esac
    if [ $engine = 1 ]; then
      line=$(env $envs python3 - <<'PY' 2>&1 | tail -1
import sys, os, argparse
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mcmc-symreg_amd"))
import bench
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
a = bench.engine_leg(args, ranks, chains=1, seconds=2.0)
b = bench.engine_leg(args, ranks)
print("one chain %.0f consumed/s (memo %.3f, discarded %.3f); eight chains %.0f (memo %.3f, discarded %.3f)" % (
    a["value"], a.get("memo_answered_fraction_of_generated", 0), a["discarded_fraction"],
    b["value"], b.get("memo_answered_fraction_of_generated", 0), b["discarded_fraction"]))
PY
)
    else
      line=$(env $envs python3 bench.py --cpu-sample 0 --extras 0 $args 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); r = d['roofline']
    print('%.0f proposals/s, %.2f us per step, row pass %.1f us (in region %.1f), frac %.3f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['kernel_us_in_timed_region'], r['frac']))
except Exception as e:
    print('FAILED', e)")
    fi
    echo "round $r  $label  [$envs${args:+ -- $args}]  $line" | tee -a "$out"
  done
done
