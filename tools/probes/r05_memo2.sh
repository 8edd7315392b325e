timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_config4.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension\|trace " | tail -4
python - <<'PY'
import sys, os, argparse, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mcmc-symreg_amd"))
import bench
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
for defer in ("1", "0", "1", "0"):
    os.environ["BSR_ENGINE_DEFER"] = defer
    a = bench.engine_leg(args, ranks, chains=1, batch=32, seconds=2.0)
    print("engine defer", defer, "c2_native_engine %.0f (memo share %.3f, discarded %.3f)" % (a["value"], a["memo_answered_fraction_of_generated"], a["discarded_fraction"]), flush=True)
PY
