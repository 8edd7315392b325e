mkdir -p gpurun_out/r05b
for v in fuse nofuse fuse2 nofuse2 noasm fuse3; do
  unset BSR_TILE_ASM BSR_FUSE_SOLVE
  case $v in nofuse*) export BSR_FUSE_SOLVE=0;; noasm*) export BSR_TILE_ASM=0;; esac
  python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05b/bench_$v.json 2> gpurun_out/r05b/bench_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05b/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"]), round(d["ms_per_step"]*1000,2), round(d["roofline"]["kernel_us"],1), round(d["roofline"]["kernel_us_in_timed_region"],1))
PY
done
unset BSR_TILE_ASM BSR_FUSE_SOLVE
for dd in 7 8; do python bench.py --cpu-sample 0 --extras 0 --depth $dd > gpurun_out/r05b/bench_fuse_d$dd.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r05b/bench_fuse_d$dd.json").read().strip().splitlines()[-1])
print("fuse depth $dd", round(d["value"]), round(d["ms_per_step"]*1000,2))
PY
done
BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep -A2 "host cost" | tail -3
python tools/host_profile.py 2>&1 | tail -4
