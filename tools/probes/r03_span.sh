p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,2), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1))"; }
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "variants or negation or linear or any_batch" 2>&1 | grep -E "^E|passed|failed" | head
for rep in 1 2; do
python bench.py --extras 0 --cpu-sample 0 | p c2
python bench.py --extras 0 --cpu-sample 0 --workload c3 | p c3
BSR_SELFDUP=0 python bench.py --extras 0 --cpu-sample 0 --workload c3 | p c3_selfdup0
done
python bench.py --extras 0 --cpu-sample 0 --chains 8 --batch 32 | p c4
BSR_HOST_PROF=1 python bench.py --extras 0 --cpu-sample 0 2>&1 | grep -i "stage\|issuing" | head -4
echo "stress: $(timeout 300 python tools/probes/ctx_sequence_stress.py R NB=2 2>&1 | grep -c False) bad"
