timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_regimes.py tests/test_gpu_tile_asm.py tests/test_gpu_ctx_sequence.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -5
mkdir -p gpurun_out/r05d
for v in a b; do python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05d/bench_$v.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r05d/bench_$v.json").read().strip().splitlines()[-1])
print("c2 $v", round(d["value"]), round(d["ms_per_step"]*1000,2), d["regions"]["spread"])
PY
done
python bench.py --cpu-sample 0 --extras 0 --workload c3 > gpurun_out/r05d/bench_c3.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r05d/bench_c3.json").read().strip().splitlines()[-1])
print("c3", round(d["value"]), round(d["ms_per_step"]*1000,2), d["roofline"]["kernel_us"], d["config"]["geometry"])
PY
BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep -A2 "host cost" | tail -3
