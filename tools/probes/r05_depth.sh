mkdir -p gpurun_out/r05a
for d in 6 7 8; do for v in 1 0; do
  BSR_TILE_ASM=$v python bench.py --cpu-sample 0 --extras 0 --depth $d > gpurun_out/r05a/bench_d${d}_asm$v.json 2> gpurun_out/r05a/bench_d${d}_asm$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05a/bench_d${d}_asm$v.json").read().strip().splitlines()[-1])
print("depth $d asm $v", round(d["value"]), round(d["ms_per_step"]*1000,2), round(d["roofline"]["kernel_us_in_timed_region"],1))
PY
done; done
