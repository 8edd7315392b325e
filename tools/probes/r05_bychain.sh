mkdir -p gpurun_out/r05g
timeout 2400 python -m pytest tests/test_gpu_regimes.py tests/test_gpu_edges.py tests/test_gpu_kernels.py tests/test_gpu_chain.py tests/test_gpu_config4.py tests/test_host_driver.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -6
for cfg in "X=1" "BSR_TILE_BY_CHAIN=0" "X=1" "BSR_TILE_BY_CHAIN=0"; do
env $cfg timeout 600 python bench.py --workload c4 --cpu-sample 0 --extras 0 --min-time 0.7 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
    print("$cfg c4", round(d["value"]), round(d["ms_per_step"]*1000,2), "row pass", round(d["roofline"]["kernel_us"],1), d["config"]["geometry"])
except Exception as e:
    print("$cfg failed", e); print(open("gpurun_out/r05g/x.err").read()[-300:])
PY
done
