run() { python bench.py --steps 100 --warmup 10 --cpu-sample 0 --workload ${WL:-c2} > gpurun_out/x.json; python -c "
import json,sys; d=json.load(open('gpurun_out/x.json')); print('$1', round(d['value']), round(d['ms_per_step']*1000,1), round(d['roofline']['kernel_us'],1))"; }
run u2_w5_c2; WL=c5 run u2_w5_c5
BSR_P1_U=4 BSR_WGS_PER_CU=4 run u4_w4_c2; BSR_P1_U=4 BSR_WGS_PER_CU=4 WL=c5 run u4_w4_c5
BSR_P1_U=4 BSR_WGS_PER_CU=5 run u4_w5_c2; BSR_P1_U=4 BSR_WGS_PER_CU=5 WL=c5 run u4_w5_c5
BSR_P1_U=2 BSR_WGS_PER_CU=4 run u2_w4_c2; BSR_WGS_PER_CU=4 WL=c5 run u2_w4_c5
BSR_P1_U=8 BSR_WGS_PER_CU=4 run u8_w4_c2; BSR_P1_U=8 BSR_WGS_PER_CU=4 WL=c5 run u8_w4_c5
