# Interleaved A/B of two BUILDS of the library in one box (boxes differ by +-4 %, a build-time change by less):
#
#   1. build variant A, `cp mcmc-symreg_amd/bsr/libbsr_hip.so tools/probes/_ab/libbsr_a.so`; the same for B
#      (tools/probes/_ab/ travels to the GPU box with the snapshot; delete it afterwards -- it is not tracked)
#   2. /usr/local/graft/bin/gpurun -- 'bash tools/probes/ab_builds.sh a b [bench args...]'
#
# BSR_LIB_PATH (bsr/_lib.py) picks the library a process loads; three rounds, A and B alternating.
a=$1; b=$2; shift 2
args="${@:---workload c5 --steps 20 --warmup 3 --cpu-sample 0 --extras 0}"
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in $a $b; do
    BSR_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/_ab/libbsr_$v.so python bench.py $args 2>/dev/null | tail -1 |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), 'per s,', round(d['ms_per_step']*1e3, 2), 'us per step, row pass', round(d['roofline']['kernel_us'], 1), 'us, frac', round(d['roofline']['frac'], 3))"
  done
done
