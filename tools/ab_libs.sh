for i in 1 2; do
for v in hip prev; do
  BSR_LIB_PATH=$GRAFT_REPO_ROOT/mcmc-symreg_amd/bsr/libbsr_$v.so python bench.py --steps 200 --warmup 20 --extras 0 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value']), d['roofline']['kernel_us'], d['roofline']['kernel_us_in_timed_region'])"
done; done
