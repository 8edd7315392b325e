mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_tile_asm.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -8
for v in asm noasm asm2 noasm2; do
  case $v in noasm*) export BSR_TILE_ASM=0;; *) unset BSR_TILE_ASM;; esac
  python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05a/bench_$v.json 2> gpurun_out/r05a/bench_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05a/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"]), d["ms_per_step"], d["roofline"]["kernel_us"])
PY
done
unset BSR_TILE_ASM
BSR_ASM_STATS=1 python bench.py --cpu-sample 0 --extras 0 --steps 100 --min-time 0 2>&1 >/dev/null | grep "tile asm" | tail -1
BSR_TILE_STAMPS=1 python tools/tile_stamps.py --workload c2 --batch 64 > gpurun_out/r05a/stamps_asm.txt 2>&1
head -16 gpurun_out/r05a/stamps_asm.txt
