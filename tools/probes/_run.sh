mkdir -p gpurun_out/r06g
python -m pytest tests/test_gpu_stream.py tests/test_gpu_fullsize.py tests/test_gpu_regimes.py -m gpu -q -s -k "f32 or fp32 or regime" > gpurun_out/r06g/tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|f32 storage|fp32 vs oracle|^E  " gpurun_out/r06g/tests.log | tail -30
for dt in f32 f64; do python bench.py --workload c5 --dtype $dt --cpu-sample 0 --extras 0 2>gpurun_out/r06g/c5_$dt.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('c5 $dt', round(d['value']), round(d['ms_per_step']*1e3,2), 'us/step; kernel', round(r['kernel_us'],1), 'frac', round(r['frac'],3), d['config']['geometry'])"; tail -2 gpurun_out/r06g/c5_$dt.err; done
for dm in 8 24 32; do BSR_DERIVED_MAX=$dm python bench.py --workload c5 --dtype f32 --cpu-sample 0 --extras 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('c5 f32 derived_max $dm', round(d['value']), round(d['ms_per_step']*1e3,2), 'us/step; kernel', round(r['kernel_us'],1))"; done
bash tools/engine_cpus.sh gpurun_out/r06g/engine_cpus.txt > /dev/null 2>&1; cat gpurun_out/r06g/engine_cpus.txt
