mkdir -p gpurun_out/r05g
timeout 1500 python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_ctx_sequence.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tail -3
run() {  # label, depth, rows, env...
  local label=$1; local depth=$2; local rows=$3; shift; shift; shift
  env "$@" timeout 600 python bench.py --cpu-sample 0 --extras 0 --rows $rows --min-time 0.7 --depth $depth > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
    print("rows $rows depth $depth $label", round(d["value"]), round(d["ms_per_step"]*1000,2), d["dispatch"])
except Exception as e:
    print("$label failed", e); print(open("gpurun_out/r05g/x.err").read()[-300:])
PY
}
for rows in 0 2048; do
for g in 1 2 3 4; do
run "group $g" 8 $rows BSR_AQL_GROUP=$g
done
run "group 2 threads1" 8 $rows BSR_AQL_GROUP=2 BSR_SUBMIT_THREADS=1
run "group 4 threads1" 8 $rows BSR_AQL_GROUP=4 BSR_SUBMIT_THREADS=1
run "group 2 depth6" 6 $rows BSR_AQL_GROUP=2
run "group 2 q2" 8 $rows BSR_AQL_GROUP=2 BSR_AQL_QUEUES=2
run "group 4 q2 threads1" 8 $rows BSR_AQL_GROUP=4 BSR_AQL_QUEUES=2 BSR_SUBMIT_THREADS=1
done
