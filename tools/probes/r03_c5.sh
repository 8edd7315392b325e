p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1), 'frac', round(r['frac'],3), d.get('verified'))"; }
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_fullsize.py -x -q -m gpu -k "variants or any_batch or fullsize or fp32" 2>&1 | tail -5
python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5
BSR_TILE_RING=2 python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5_ring2
BSR_TILE_RING=3 python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5_ring3
BSR_DERIVED=0 python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5_noderived
BSR_TILE_T=4 python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5_T4
BSR_TILE_T=3 python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5_T3
BSR_DERIVED_MAX=32 python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5_d32
BSR_DERIVED_MAX=8 python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5_d8
python bench.py --extras 0 --cpu-sample 0 --workload c5 --dtype f32 | p c5_f32
python bench.py --extras 0 --cpu-sample 0 --workload c5 --depth 1 | p c5_depth1
python bench.py --extras 0 --cpu-sample 0 --workload c5 --depth 2 | p c5_depth2
BSR_TILE_STAMPS=1 BSR_STREAM_STATS=1 python tools/tile_stamps.py --workload c5 2>&1 | grep -E "geometry|stream stats|first chunk|tape group|wave end"
