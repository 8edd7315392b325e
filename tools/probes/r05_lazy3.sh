for cfg in "BSR_LAZY_TAIL=0" "BSR_LAZY_TAIL=1" "BSR_LAZY_TAIL=2" "BSR_LAZY_TAIL=1 BSR_LAZY_TAIL_DEPTH=4" "BSR_LAZY_TAIL=0" "BSR_LAZY_TAIL=1"; do
env $cfg python - <<'PY'
import sys, os, argparse, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mcmc-symreg_amd"))
import bench
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("BSR_LAZY"))
a = bench.engine_leg(args, ranks, chains=1, batch=32, seconds=2.0)
b = bench.engine_leg(args, ranks)
print(tag, "c2_native_engine %.0f   c4_native_engine %.0f" % (a["value"], b["value"]), flush=True)
PY
done
mkdir -p gpurun_out/r05g
for cfg in "BSR_LAZY_TAIL=0" "BSR_LAZY_TAIL=1"; do
env $cfg timeout 600 python bench.py --cpu-sample 0 --extras 0 --min-time 0.7 --depth 8 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
print("headline $cfg", round(d["value"]), round(d["ms_per_step"]*1000,2), d["dispatch"])
PY
done
