p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1))"; }
for v in 0 1; do
HIP_FORCE_DEV_KERNARG=$v BSR_TILE_STAMPS=1 python tools/tile_stamps.py 2>&1 | grep -E "geometry|stage first|wave start|wave end"
HIP_FORCE_DEV_KERNARG=$v python bench.py --extras 0 --cpu-sample 0 | p kernarg$v
HIP_FORCE_DEV_KERNARG=$v python bench.py --extras 0 --cpu-sample 0 --workload c3 | p c3_kernarg$v
done
