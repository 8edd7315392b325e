timeout 1800 python -m pytest tests/test_gpu_regimes.py tests/test_gpu_config4.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -5
for i in 1 2; do
python - <<'PY' 2>&1 | tail -1
import sys, os, argparse, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mcmc-symreg_amd"))
import bench
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
a = bench.engine_leg(args, ranks, chains=1, batch=32, seconds=2.0)
b = bench.engine_leg(args, ranks)
print("c2_native_engine %.0f  c4_native_engine %.0f discarded %.3f" % (a["value"], b["value"], b["discarded_fraction"]), flush=True)
PY
done
