# Row-pass variants on the bench workload (run on the GPU box).  Prints: tag, proposals/s, us/step, kernel us (isolated)
run() { python bench.py --steps 60 --warmup 5 --cpu-sample 0 --batch ${B:-64} --depth ${D:-3} --workload ${WL:-c2} > gpurun_out/x.json; python -c "
import json,sys; d=json.load(open('gpurun_out/x.json')); print('$1', round(d['value']), round(d['ms_per_step']*1000,1), round(d['roofline']['kernel_us'],1))"; }
BSR_P1_U=2 run glb_u2
BSR_P1_U=4 run glb_u4
BSR_RB_ROWS=1024 run glb_u2_rb1024
BSR_RB_ROWS=2048 run glb_u2_rb2048
BSR_RB_ROWS=1024 BSR_P1_U=4 run glb_u4_rb1024
BSR_RB_ROWS=256 run glb_u2_rb256
BSR_TARGET_WGS=4096 run glb_u2_wgs4096
BSR_TARGET_WGS=4096 BSR_RB_ROWS=1024 run glb_u2_rb1024_wgs4096
BSR_TARGET_WGS=1024 run glb_u2_wgs1024
BSR_NO_LDS=0 BSR_P1_U=2 run lds_u2
WL=c5 run c5_default
WL=c5 BSR_RB_ROWS=1024 run c5_rb1024
WL=c5 BSR_RB_ROWS=2048 run c5_rb2048
WL=c5 BSR_P1_U=4 run c5_u4
WL=c3 run c3_default
WL=c3 BSR_RB_ROWS=1024 run c3_rb1024
