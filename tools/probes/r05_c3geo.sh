mkdir -p gpurun_out/r05g
for cfg in "X=1" "BSR_TILE_BPS=7" "BSR_TILE_BPS=6" "BSR_TILE_BPS=5" "BSR_AUX_CUS=0" "BSR_AUX_CUS=128" "X=1" "BSR_TILE_BPS=7"; do
env $cfg timeout 600 python bench.py --workload c3 --cpu-sample 0 --extras 0 --min-time 0.7 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
    print("$cfg c3", round(d["value"]), round(d["ms_per_step"]*1000,2), "row pass", round(d["roofline"]["kernel_us"],1), d["config"]["geometry"])
except Exception as e:
    print("$cfg failed", e); print(open("gpurun_out/r05g/x.err").read()[-300:])
PY
done
