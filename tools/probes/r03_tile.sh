# quick regression + timing of the row pass (GPU box)
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1))"; }
timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "variants or any_batch" 2>&1 | tail -12
for rep in 1 2; do
python bench.py --extras 0 --cpu-sample 0 | p base
BSR_TILE_T=2 python bench.py --extras 0 --cpu-sample 0 | p T2
BSR_AUX_CUS=0 python bench.py --extras 0 --cpu-sample 0 | p aux0
BSR_DERIVED_MAX=16 python bench.py --extras 0 --cpu-sample 0 | p dmax16
done
python bench.py --extras 0 --cpu-sample 0 --workload c3 | p c3
python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5
BSR_TILE_T=4 python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5_T4
BSR_TILE_T=1 python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5_T1
python bench.py --extras 0 --cpu-sample 0 --chains 8 --batch 32 | p c4
BSR_TILE_STAMPS=1 python tools/tile_stamps.py 2>&1 | grep -E "geometry|lifetime|stage first|all chunks|imbalance|wave end|per workgroup|reduce"
