mkdir -p gpurun_out/r05g
for lib in "" sw2 sw1 "" sw2 sw1; do
  if [ -n "$lib" ]; then export BSR_LIB_PATH=$PWD/mcmc-symreg_amd/bsr/libbsr_hip_$lib.so; else unset BSR_LIB_PATH; fi
  for w in c2 c3; do
  timeout 600 python bench.py --workload $w --cpu-sample 0 --extras 0 --min-time 0.7 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
print("solve waves '${lib:-4}' $w", round(d["value"]), round(d["ms_per_step"]*1000,2))
PY
  done
done
