p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1))"; }
for rep in 1 2; do
python bench.py --extras 0 --cpu-sample 0 | p c2
BSR_TILE_LONG=0 python bench.py --extras 0 --cpu-sample 0 | p c2_long0
done
python bench.py --extras 0 --cpu-sample 0 --workload c3 | p c3
python bench.py --extras 0 --cpu-sample 0 --chains 8 --batch 32 | p c4
python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5
python bench.py --extras 0 --cpu-sample 0 --depth 1 | p c2_depth1
BSR_TILE_LONG=0 python bench.py --extras 0 --cpu-sample 0 --depth 1 | p c2_depth1_long0
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/pytest_gpu.log | grep -E "passed|failed" | tail -3
