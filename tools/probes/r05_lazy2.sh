mkdir -p gpurun_out/r05g
timeout 2400 python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_ctx_sequence.py tests/test_gpu_kernels.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r05g/pytest.txt 2>&1; tail -5 gpurun_out/r05g/pytest.txt
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
run "lazy" 8 $rows BSR_LAZY_TAIL=1
run "eager" 8 $rows BSR_LAZY_TAIL=0
run "lazy" 6 $rows BSR_LAZY_TAIL=1
run "lazy threads1" 8 $rows BSR_LAZY_TAIL=1 BSR_SUBMIT_THREADS=1
run "lazy threads3" 8 $rows BSR_LAZY_TAIL=1 BSR_SUBMIT_THREADS=3
done
run "lazy c3" 8 0 BSR_LAZY_TAIL=1
