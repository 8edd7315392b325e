mkdir -p gpurun_out/r05g
for bps in 8 5 6 7 10 8; do
BSR_TILE_BPS=$bps timeout 600 python bench.py --cpu-sample 0 --extras 0 --min-time 0.7 --depth 8 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
    print("bps $bps", round(d["value"]), round(d["ms_per_step"]*1000,2), "row pass us", round(d["roofline"]["kernel_us"],2), d["config"]["geometry"])
except Exception as e:
    print("bps $bps failed", e); print(open("gpurun_out/r05g/x.err").read()[-300:])
PY
done
for rows in 50000 25000; do
for cfg in "BSR_AQL=1" "BSR_AQL=0"; do
env $cfg timeout 600 python bench.py --cpu-sample 0 --extras 0 --rows $rows --min-time 0.7 --depth 8 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
print("rows $rows $cfg", round(d["value"]), round(d["ms_per_step"]*1000,2), "row pass us", round(d["roofline"]["kernel_us"],2), d["config"]["geometry"])
PY
done
done
