mkdir -p gpurun_out/r05g
for cfg in "BSR_LAZY_TAIL=0" "BSR_LAZY_TAIL=1" "BSR_LAZY_TAIL=0" "BSR_LAZY_TAIL=1"; do
for w in c3 c5 c4; do
env $cfg timeout 600 python bench.py --workload $w --cpu-sample 0 --extras 0 --min-time 0.7 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
    print("$cfg $w", round(d["value"]), round(d["ms_per_step"]*1000,2), d["dispatch"])
except Exception as e:
    print("$cfg $w failed", e); print(open("gpurun_out/r05g/x.err").read()[-300:])
PY
done
done
