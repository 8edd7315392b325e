timeout 1500 python -m pytest tests/test_gpu_ctx_sequence.py tests/test_gpu_kernels.py tests/test_gpu_tile_asm.py -x -q -m gpu 2>&1 | grep "passed\|failed" | tail -3
mkdir -p gpurun_out/r05e
for v in w1 w0 w1b w0b; do
  case $v in w0*) export BSR_DONE_WORD=0;; *) unset BSR_DONE_WORD;; esac
  python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05e/bench_$v.json 2>gpurun_out/r05e/bench_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05e/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"]), round(d["ms_per_step"]*1000,2), d["regions"]["spread"])
PY
done
unset BSR_DONE_WORD
for v in 1 0; do BSR_DONE_WORD=$v python bench.py --cpu-sample 0 --extras 0 --rows 2048 > gpurun_out/r05e/rows_$v.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r05e/rows_$v.json").read().strip().splitlines()[-1])
print("rows2048 done_word=$v", round(d["value"]), round(d["ms_per_step"]*1000,2))
PY
done
BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep -A2 "host cost" | tail -3
