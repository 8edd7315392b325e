p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1))"; }
for rep in 1 2; do
python bench.py --extras 0 --cpu-sample 0 | p base
BSR_TILE_SUB=2 python bench.py --extras 0 --cpu-sample 0 | p sub2
BSR_AUX_CUS=0 python bench.py --extras 0 --cpu-sample 0 | p aux0
BSR_AUX_CUS=32 python bench.py --extras 0 --cpu-sample 0 | p aux32
BSR_AUX_CUS=96 python bench.py --extras 0 --cpu-sample 0 | p aux96
BSR_AUX_CUS=128 python bench.py --extras 0 --cpu-sample 0 | p aux128
python bench.py --extras 0 --cpu-sample 0 --depth 8 | p depth8
python bench.py --extras 0 --cpu-sample 0 --depth 4 | p depth4
BSR_DERIVED_MAX=16 python bench.py --extras 0 --cpu-sample 0 | p dmax16
BSR_DERIVED_MAX=4 python bench.py --extras 0 --cpu-sample 0 | p dmax4
done
python bench.py --extras 0 --cpu-sample 0 --workload c3 | p c3
python bench.py --extras 0 --cpu-sample 0 --workload c5 | p c5
python bench.py --extras 0 --cpu-sample 0 --chains 8 --batch 32 | p c4
