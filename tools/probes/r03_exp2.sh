p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1))"; }
python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "variants or any_batch" 2>&1 | tail -5
for rep in 1 2; do
for sc in 0 45 60 80 110; do BSR_TILE_SPLIT_COST=$sc python bench.py --extras 0 --cpu-sample 0 | p split$sc; done
done
BSR_TILE_STAMPS=1 python tools/tile_stamps.py 2>&1 | grep -E "geometry|lifetime|stage first|all chunks|imbalance|wave end|per workgroup"
BSR_TILE_SPLIT_COST=0 BSR_TILE_STAMPS=1 python tools/tile_stamps.py 2>&1 | grep -E "geometry|lifetime|stage first|all chunks|imbalance|wave end|per workgroup"
