timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_chain.py tests/test_gpu_config4.py tests/test_gpu_regimes.py tests/test_gpu_edges.py tests/test_gpu_host_driver.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|assert" | tail -6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c2 c3; do
rm -rf gpurun_out/ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --workload $w --steps 200 --warmup 20 --cpu-sample 0 --extras 0 --min-time 0 > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/ks/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("k_solve") or "k_tile" in r["Name"]:
        print("$w", r["Name"][:40], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3, 2))
PY
done
rm -rf gpurun_out/ks
for w in c2 c3; do
timeout 600 python bench.py --workload $w --cpu-sample 0 --extras 0 --min-time 0.7 --depth 8 > gpurun_out/r05g/x.json 2>gpurun_out/r05g/x.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05g/x.json").read().strip().splitlines()[-1])
print("$w", round(d["value"]), round(d["ms_per_step"]*1000,2))
PY
done
