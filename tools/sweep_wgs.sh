run() { python bench.py --steps 60 --warmup 5 --cpu-sample 0 --batch ${B:-64} --depth ${D:-3} --workload ${WL:-c2} > gpurun_out/x.json; python -c "
import json,sys; d=json.load(open('gpurun_out/x.json')); print('$1', round(d['value']), round(d['ms_per_step']*1000,1), round(d['roofline']['kernel_us'],1))"; }
for w in 1000 1200 1280 1400 1600 2048 2500 3200; do BSR_TARGET_WGS=$w run wgs$w; done
for w in 1200 2048; do BSR_TARGET_WGS=$w BSR_RB_ROWS=256 run rb256_wgs$w; done
for w in 1200 2048; do BSR_TARGET_WGS=$w BSR_RB_ROWS=1024 run rb1024_wgs$w; done
