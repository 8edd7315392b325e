#!/bin/bash
# rocprofv3 kernel stats of the bench command in its pipelined default (what the judge reads) for one workload.
# usage: bash tools/r06_stats.sh <tag> <workload> [bench args...]
tag=$1; wl=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python3 bench.py --workload $wl --cpu-sample 0 --extras 0 "$@" > $out/bench_${wl}_under_rocprof.json 2> $out/stats_$wl.err
f=$(ls $out/stats_$wl/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats_bench_${wl}.csv && head -8 $out/kernel_stats_bench_${wl}.csv | cut -c1-200
rm -rf $out/stats_$wl
