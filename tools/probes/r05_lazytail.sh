timeout 2400 python -m pytest tests/test_gpu_ctx_sequence.py tests/test_gpu_kernels.py tests/test_gpu_tile_asm.py tests/test_gpu_edges.py tests/test_gpu_chain.py tests/test_gpu_config4.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tail -5
mkdir -p gpurun_out/r05f
for v in l1 l0 l1b l0b; do
  case $v in l0*) export BSR_LAZY_TAIL=0;; *) unset BSR_LAZY_TAIL;; esac
  python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05f/bench_$v.json 2>gpurun_out/r05f/bench_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05f/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"]), round(d["ms_per_step"]*1000,2), d["regions"]["spread"])
PY
done
unset BSR_LAZY_TAIL
for v in 1 0; do BSR_LAZY_TAIL=$v python bench.py --cpu-sample 0 --extras 0 --rows 2048 > gpurun_out/r05f/rows_$v.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r05f/rows_$v.json").read().strip().splitlines()[-1])
print("rows2048 lazy=$v", round(d["value"]), round(d["ms_per_step"]*1000,2))
PY
done
BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep -A2 "host cost" | tail -3
