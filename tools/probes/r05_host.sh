mkdir -p gpurun_out/r05a
for t in 1 2 3 4; do
  BSR_SUBMIT_THREADS=$t python bench.py --cpu-sample 0 --extras 0 --rows 2048 > gpurun_out/r05a/bench_rows2048_t$t.json 2> gpurun_out/r05a/bench_rows2048_t$t.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05a/bench_rows2048_t$t.json").read().strip().splitlines()[-1])
print("rows 2048 submit threads $t", round(d["value"]), round(d["ms_per_step"]*1000,2))
PY
done
for t in 3 4; do
  BSR_SUBMIT_THREADS=$t python bench.py --cpu-sample 0 --extras 0 --depth 7 > gpurun_out/r05a/bench_full_t$t.json 2> gpurun_out/r05a/bench_full_t$t.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05a/bench_full_t$t.json").read().strip().splitlines()[-1])
print("full N submit threads $t depth 7", round(d["value"]), round(d["ms_per_step"]*1000,2))
PY
done
BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 --rows 2048 2>&1 >/dev/null | grep "host cost" | tail -1
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
