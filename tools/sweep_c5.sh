run() { python bench.py --steps 60 --warmup 5 --cpu-sample 0 --workload ${WL:-c5} --batch ${B:-64} > gpurun_out/x.json; python -c "
import json,sys; d=json.load(open('gpurun_out/x.json')); print('$1', round(d['value']), round(d['ms_per_step']*1000,1), round(d['roofline']['kernel_us'],1), round(d['roofline']['frac'],3))"; }
run c5_rb1024; BSR_RB_ROWS=2048 run c5_rb2048; BSR_RB_ROWS=4096 run c5_rb4096; BSR_RB_ROWS=512 run c5_rb512
B=16 run c5_B16; B=32 run c5_B32; B=128 run c5_B128
