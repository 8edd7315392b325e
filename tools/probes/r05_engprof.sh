BSR_ENGINE_PROF=1 BSR_HOST_PROF=1 python - <<'PY' 2>&1 | tail -30
import sys, os, argparse, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mcmc-symreg_amd"))
import bench
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
a = bench.engine_leg(args, ranks, chains=1, batch=32, seconds=2.0)
print({k: v for k, v in a.items() if not isinstance(v, (dict, list))})
PY
