#!/bin/bash
# interleaved A/B of BSR_AUX_CUS with the bench's default step counts (and with long ones)
for rep in 1 2 3; do
  for v in ${AUX_LIST:-0 64}; do
    for st in "--steps 200 --warmup 20" "--steps 2000 --warmup 200"; do
      r=$(BSR_AUX_CUS=$v python bench.py $st --cpu-sample 0 --extras 0 2>/dev/null | tail -1 |
          python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M/s %.2f us  solo %.1f in-region %.1f' % (d['value']/1e6, d['ms_per_step']*1000, d['roofline']['kernel_us'], d['roofline']['kernel_us_in_timed_region']))")
      echo "rep=$rep AUX=$v [$st] $r"
    done
  done
done
