p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1), d.get('verified'))"; }
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "variants or any_batch or negation" 2>&1 | grep -E "^E|passed|failed" | head -20
for rep in 1 2; do
timeout 300 python bench.py --extras 0 --cpu-sample 0 | p fused
BSR_FUSED_TAIL=0 timeout 300 python bench.py --extras 0 --cpu-sample 0 | p legacy
BSR_TILE_WGS=96 BSR_TILE_T=1 BSR_SUBMIT_THREADS=2 timeout 300 python bench.py --extras 0 --cpu-sample 0 | p fused_wgs96_T1_t2
BSR_FUSED_TAIL=0 BSR_TILE_WGS=96 BSR_TILE_T=1 BSR_SUBMIT_THREADS=2 timeout 300 python bench.py --extras 0 --cpu-sample 0 | p legacy_wgs96_T1_t2
done
