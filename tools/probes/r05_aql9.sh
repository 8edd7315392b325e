for cfg in "BSR_AQL=0" "BSR_AQL_GROUP=1" "BSR_AQL_GROUP=2" "BSR_AQL_GROUP=4"; do
env $cfg python - <<'PY'
import sys, os, argparse, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mcmc-symreg_amd"))
import bench
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
tag = " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("BSR_AQL"))
a = bench.engine_leg(args, ranks, chains=1, batch=32, seconds=2.0)
print(tag, "c2_native_engine %.0f (memo share %.3f)" % (a["value"], a["memo_answered_fraction_of_generated"]), flush=True)
b = bench.engine_leg(args, ranks)
print(tag, "c4_native_engine %.0f" % b["value"], flush=True)
PY
done
for rows in 50000 25000; do
for cfg in "BSR_AQL=1" "BSR_AQL=0"; do
env $cfg timeout 600 python bench.py --cpu-sample 0 --extras 0 --rows $rows --min-time 0.7 --depth 8 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
print("rows $rows $cfg", round(d["value"]), round(d["ms_per_step"]*1000,2), "row pass us", round(d["roofline"]["kernel_us"],2))
PY
done
done
