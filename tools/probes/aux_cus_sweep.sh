#!/bin/bash
# A/B of BSR_AUX_CUS (row pass sized for n_cu - n CUs, the rest left to the small kernels), interleaved repetitions
# in one box: boxes differ by more than the effect.
for rep in ${REPS:-1 2 3}; do
  for w in c2 c3; do
    for v in ${AUX_LIST:-0 8 32 64}; do
      r=$(BSR_AUX_CUS=$v python bench.py --workload $w --steps 2000 --warmup 200 --cpu-sample 0 --extras 0 2>&1 | tail -1 |
          python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M/s %.2f us' % (d['value']/1e6, d['ms_per_step']*1000))")
      echo "rep=$rep W=$w AUX=$v $r"
    done
  done
done
