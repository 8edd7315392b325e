p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1))"; }
for rep in 1 2; do
python bench.py --extras 0 --cpu-sample 0 2>/dev/null| p base
BSR_X_SKIP_RESID=1 python bench.py --extras 0 --cpu-sample 0 2>/dev/null| p skipresid
BSR_TILE_WGS=96 BSR_TILE_T=1 BSR_SUBMIT_THREADS=2 python bench.py --extras 0 --cpu-sample 0 2>/dev/null | p wgs96_T1_t2
BSR_X_SKIP_RESID=1 BSR_TILE_WGS=96 BSR_TILE_T=1 BSR_SUBMIT_THREADS=2 python bench.py --extras 0 --cpu-sample 0 2>/dev/null | p skipresid_wgs96_T1_t2
done
