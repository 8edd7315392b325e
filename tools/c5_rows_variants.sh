# Rows-per-lane variants of the work-queue pass at N=1M (builds with a lower occupancy floor: -DBSR_ROWS_MIN_WAVES=3 / 2
# as mcmc-symreg_amd/bsr/libbsr_mw3.so / libbsr_mw2.so).  usage: bash tools/c5_rows_variants.sh
run() { python bench.py --workload c5 --steps 100 --warmup 10 --extras 0 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1: value', round(d['value']), 'row pass %.1f / %.1f us' % (d['roofline']['kernel_us'], d['roofline']['kernel_us_in_timed_region']), 'frac %.3f' % d['roofline']['frac'])"; }
for i in 1 2; do
run "U=2, 5 waves/SIMD (default)"
BSR_LIB_PATH=$GRAFT_REPO_ROOT/mcmc-symreg_amd/bsr/libbsr_mw3.so BSR_P1_U=4 BSR_WGS_PER_CU=3 run "U=4, 3 waves/SIMD"
BSR_LIB_PATH=$GRAFT_REPO_ROOT/mcmc-symreg_amd/bsr/libbsr_mw3.so BSR_P1_U=2 BSR_WGS_PER_CU=5 run "U=2, floor 3"
BSR_LIB_PATH=$GRAFT_REPO_ROOT/mcmc-symreg_amd/bsr/libbsr_mw2.so BSR_P1_U=8 BSR_WGS_PER_CU=2 run "U=8, 2 waves/SIMD"
BSR_LIB_PATH=$GRAFT_REPO_ROOT/mcmc-symreg_amd/bsr/libbsr_mw2.so BSR_P1_U=4 BSR_WGS_PER_CU=3 run "U=4, floor 2"
done
