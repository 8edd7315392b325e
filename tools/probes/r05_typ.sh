for cfg in "X=1" "BSR_TYPICAL_CHAINS=2 BSR_TYPICAL_BATCH=64" "X=1" "BSR_TYPICAL_CHAINS=2 BSR_TYPICAL_BATCH=64"; do
env $cfg python - <<'PY' 2>&1 | tail -1
import sys, os, argparse, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mcmc-symreg_amd"))
import bench
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("BSR_TYPICAL"))
b = bench.engine_leg(args, ranks)
print(tag or "default", "c4_native_engine %.0f discarded %.3f" % (b["value"], b["discarded_fraction"]), flush=True)
PY
done
