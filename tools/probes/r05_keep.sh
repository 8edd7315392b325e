mkdir -p gpurun_out/r05g
timeout 1800 python -m pytest tests/test_gpu_tile_asm.py tests/test_gpu_kernels.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tail -3
for lib in "" prev "" prev "" prev; do
  if [ -n "$lib" ]; then export BSR_LIB_PATH=$PWD/mcmc-symreg_amd/bsr/libbsr_hip_$lib.so; else unset BSR_LIB_PATH; fi
  timeout 600 python bench.py --cpu-sample 0 --extras 0 --min-time 0.7 --depth 8 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
print("lib '${lib:-new}'", round(d["value"]), round(d["ms_per_step"]*1000,2), "row pass us", round(d["roofline"]["kernel_us"],2))
PY
done
