p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1))"; }
timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "variants or any_batch" 2>&1 | grep -E "^E|passed|failed" | head -5
for rep in 1 2; do
for b in 8 6 5 4 10 12; do BSR_TILE_LONG=$b python bench.py --extras 0 --cpu-sample 0 | p c2_bps$b; done
done
python bench.py --extras 0 --cpu-sample 0 --workload c3 | p c3
