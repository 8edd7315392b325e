p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'in_region', round(r['kernel_us_in_timed_region'],1))"; }
for q in 4 6 8; do for dep in 6 8; do GPU_MAX_HW_QUEUES=$q python bench.py --extras 0 --cpu-sample 0 --depth $dep | p "queues$q depth$dep"; done; done
BSR_SUBMIT_THREADS=3 python bench.py --extras 0 --cpu-sample 0 | p threads3
BSR_SUBMIT_THREADS=1 python bench.py --extras 0 --cpu-sample 0 | p threads1
