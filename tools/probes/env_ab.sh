#!/bin/bash
# usage: SETS="A=1 'A=1 B=2' ..." bash tools/probes/env_ab.sh  -- interleaved A/B of whole environment settings ("-" = none)
IFS=';' read -ra sets <<< "$SETS"
for rep in ${REPS:-1 2}; do
  for w in ${WORKLOADS:-c2}; do
    for s in "${sets[@]}"; do
      e="$s"; [ "$s" = "-" ] && e=""
      r=$(env $e python bench.py --workload $w ${BENCH_ARGS:-} --steps 2000 --warmup 200 --cpu-sample 0 --extras 0 2>/dev/null | tail -1 |
          python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M/s %.2f us  solo %.1f' % (d['value']/1e6, d['ms_per_step']*1000, d['roofline']['kernel_us']))")
      echo "rep=$rep W=$w [$s] $r"
    done
  done
done
