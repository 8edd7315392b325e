mkdir -p gpurun_out/r05g
run() {  # label, depth, env...
  local label=$1; local depth=$2; shift; shift
  env "$@" timeout 600 python bench.py --cpu-sample 0 --extras 0 --rows 2048 --min-time 0.5 --depth $depth > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
    print("rows2048 depth $depth $label", round(d["value"]), round(d["ms_per_step"]*1000,2))
except Exception as e:
    print("$label failed", e); print(open("gpurun_out/r05g/x.err").read()[-600:])
PY
}
for d in 3 4 5 6; do
run "aql q8" $d BSR_AQL_QUEUES=8
done
run "aql q8 hipq1" 6 BSR_AQL_QUEUES=8 GPU_MAX_HW_QUEUES=1
run "aql q6 hipq1" 6 BSR_AQL_QUEUES=6 GPU_MAX_HW_QUEUES=1
run "aql q6 hipq2" 6 BSR_AQL_QUEUES=6 GPU_MAX_HW_QUEUES=2
run "aql q5" 5 BSR_AQL_QUEUES=5
run "hip hipq8" 6 BSR_AQL=0 GPU_MAX_HW_QUEUES=8
run "hip hipq2" 6 BSR_AQL=0 GPU_MAX_HW_QUEUES=2
