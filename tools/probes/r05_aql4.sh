mkdir -p gpurun_out/r05g
export BSR_AQL_VERBOSE=1
timeout 900 python -m pytest tests/test_gpu_ctx_sequence.py tests/test_gpu_kernels.py tests/test_gpu_tile_asm.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tail -3
run() {  # label, depth, rows, env...
  local label=$1; local depth=$2; local rows=$3; shift; shift; shift
  env "$@" timeout 600 python bench.py --cpu-sample 0 --extras 0 --rows $rows --min-time 0.7 --depth $depth > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
    print("rows $rows depth $depth $label", round(d["value"]), round(d["ms_per_step"]*1000,2))
except Exception as e:
    print("$label failed", e); print(open("gpurun_out/r05g/x.err").read()[-600:])
PY
}
for rows in 2048 0; do
run "hip" 6 $rows BSR_AQL=0
run "aql r1+t3" 6 $rows BSR_AQL=1
run "aql r1+t3" 8 $rows BSR_AQL=1
run "aql r2+t2" 6 $rows BSR_AQL_ROW_QUEUES=2
run "aql r1+t2" 6 $rows BSR_AQL_QUEUES=3
run "aql r1+t4" 8 $rows BSR_AQL_QUEUES=5
run "aql nosplit" 6 $rows BSR_AQL_ROW_QUEUES=0
done
run "aql r1+t3 threads3" 8 0 BSR_SUBMIT_THREADS=3
