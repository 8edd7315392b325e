timeout 1800 python -m pytest tests/test_gpu_config4.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -3
for cfg in "BSR_ENGINE_GEN_HELPER=1" "BSR_ENGINE_GEN_HELPER=0" "BSR_ENGINE_GEN_HELPER=1" "BSR_ENGINE_GEN_HELPER=0"; do
env $cfg BSR_ENGINE_PROF=1 python - <<'PY' 2>&1 | tail -2
import sys, os, argparse, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mcmc-symreg_amd"))
import bench
args = argparse.Namespace(batch=0, chains=0, dtype="f64", burnin=300, rows=0)
ranks = bench.Ranks()
b = bench.engine_leg(args, ranks)
print(os.environ.get("BSR_ENGINE_GEN_HELPER"), "c4_native_engine %.0f discarded %.3f" % (b["value"], b["discarded_fraction"]), flush=True)
PY
done
