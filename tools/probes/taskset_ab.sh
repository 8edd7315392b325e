#!/bin/bash
# does confining the process to a few neighbouring CPUs steady the pipelined step?  (threads otherwise roam 256 CPUs)
lscpu | grep -i "l3\|thread(s) per core\|core(s) per socket\|numa node0" | head -5
cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list 2>/dev/null
for rep in 1 2 3; do
  for cpus in all 0-7 0-15 0-3; do
    if [ $cpus = all ]; then pre=""; else pre="taskset -c $cpus"; fi
    r=$($pre python bench.py --steps 2000 --warmup 200 --cpu-sample 0 --extras 0 2>/dev/null | tail -1 |
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M/s %.2f us' % (d['value']/1e6, d['ms_per_step']*1000))")
    echo "rep=$rep cpus=$cpus $r"
  done
done
