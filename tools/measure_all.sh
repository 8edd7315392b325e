p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'us/step', round(d['ms_per_step']*1000,1), 'kernel_us', round(r['kernel_us'],1), 'in_region', round(r['kernel_us_in_timed_region'],1), 'frac', round(r['frac'],4), 'GB/s', round(r['achieved']), 'bytes', r['algorithmic_bytes'], 'cpu', d.get('cpu_baseline',{}).get('value'))"; }
python bench.py | tee gpurun_out/bench_c2.json | p c2
python bench.py --workload c3 --cpu-sample 0 | p c3
python bench.py --workload c5 --cpu-sample 0 | p c5
python bench.py --workload c5 --dtype f32 --cpu-sample 0 | p c5_f32
python bench.py --chains 8 --batch 32 --cpu-sample 0 | p c4share
python bench.py --batch 256 --cpu-sample 0 | p c2_B256
python bench.py --depth 1 --cpu-sample 0 | p c2_sync
for a in "--chains 1 --batch 32" "--chains 1 --batch 64" "--chains 8 --batch 32" "--chains 64 --batch 16" "--K 8 --chains 8 --batch 32"; do python tools/chain_throughput.py --props 20000 $a; done
BSR_ENGINE_THREADS=0 python tools/chain_throughput.py --props 20000 --chains 8 --batch 32
python tools/chain_throughput.py --props 3000 --chains 8 --batch 32 --engine python
