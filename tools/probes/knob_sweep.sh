#!/bin/bash
# usage: KNOB=BSR_DERIVED_MAX VALUES="0 4 8" [WORKLOADS="c2 c3"] bash tools/probes/knob_sweep.sh -- interleaved A/B of one env knob
for rep in ${REPS:-1 2}; do
  for w in ${WORKLOADS:-c2 c3}; do
    for v in $VALUES; do
      r=$(env $KNOB=$v python bench.py --workload $w --steps 2000 --warmup 200 --cpu-sample 0 --extras 0 2>/dev/null | tail -1 |
          python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M/s %.2f us  solo %.1f' % (d['value']/1e6, d['ms_per_step']*1000, d['roofline']['kernel_us']))")
      echo "rep=$rep W=$w $KNOB=$v $r"
    done
  done
done
