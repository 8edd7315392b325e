#!/bin/bash
for rep in 1 2; do for w in ${WORKLOADS:-c2 c3}; do for dp in ${DEPTHS:-3 4 6 8}; do
  r=$(python bench.py --workload $w --depth $dp --steps 2000 --warmup 200 --cpu-sample 0 --extras 0 2>/dev/null | tail -1 |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M/s %.2f us  in-region %.1f' % (d['value']/1e6, d['ms_per_step']*1000, d['roofline']['kernel_us_in_timed_region']))")
  echo "rep=$rep W=$w depth=$dp $r"
done; done; done
