mkdir -p gpurun_out/r05b
for v in asm noasm asm2 noasm2 asm3 noasm3; do
  case $v in noasm*) export BSR_TILE_ASM=0;; *) unset BSR_TILE_ASM;; esac
  python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05b/bench_$v.json 2> gpurun_out/r05b/bench_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05b/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"]), round(d["ms_per_step"]*1000,2), round(d["roofline"]["kernel_us"],1), round(d["roofline"]["kernel_us_in_timed_region"],1))
PY
done
unset BSR_TILE_ASM
for v in 1 0; do
BSR_TILE_ASM=$v BSR_TILE_STAMPS=1 python tools/tile_stamps.py --workload c2 --batch 64 > gpurun_out/r05b/stamps_asm$v.txt 2>&1
echo "== stamps asm=$v"; sed -n '1p;4,5p;6p;8,12p;14p' gpurun_out/r05b/stamps_asm$v.txt
done
