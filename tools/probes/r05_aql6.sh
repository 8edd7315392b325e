mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests/test_gpu_ctx_sequence.py tests/test_gpu_kernels.py tests/test_gpu_regimes.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tail -3
run() {  # label, depth, rows, env...
  local label=$1; local depth=$2; local rows=$3; shift; shift; shift
  env "$@" timeout 600 python bench.py --cpu-sample 0 --extras 0 --rows $rows --min-time 0.7 --depth $depth > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
    print("rows $rows depth $depth $label", round(d["value"]), round(d["ms_per_step"]*1000,2), "row pass us", d["roofline"].get("launch_us"), d["roofline"].get("achieved"))
except Exception as e:
    print("$label failed", e); print(open("gpurun_out/r05g/x.err").read()[-300:])
PY
}
for rows in 0 2048; do
run "hip" 6 $rows BSR_AQL=0
run "aql" 6 $rows BSR_AQL=1
run "aql" 8 $rows BSR_AQL=1
run "aql noprofile" 8 $rows BSR_AQL_PROFILE=0
run "aql split r1" 8 $rows BSR_AQL_ROW_QUEUES=1
run "aql q3" 8 $rows BSR_AQL_QUEUES=3
run "aql threads3" 8 $rows BSR_SUBMIT_THREADS=3
run "aql threads1" 8 $rows BSR_SUBMIT_THREADS=1
done
for cfg in "BSR_AQL=1" "BSR_AQL_ROW_QUEUES=1" ; do
echo "== plain driver rows 0 $cfg"
env $cfg timeout 300 python tools/probes/two_callers.py --rows 0 --callers 2 --depth 8 2>&1 | grep "caller(s)"
done
