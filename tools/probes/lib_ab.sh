#!/bin/bash
# usage: LIBS="hip solve4 solve16" bash tools/probes/lib_ab.sh -- interleaved A/B of alternative builds (mcmc-symreg_amd/bsr/libbsr_<name>.so)
for rep in ${REPS:-1 2 3}; do for w in ${WORKLOADS:-c2 c3}; do for v in $LIBS; do
  r=$(BSR_LIB_PATH=$GRAFT_REPO_ROOT/mcmc-symreg_amd/bsr/libbsr_$v.so python bench.py --workload $w --steps 2000 --warmup 200 --cpu-sample 0 --extras 0 2>/dev/null | tail -1 |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M/s %.2f us' % (d['value']/1e6, d['ms_per_step']*1000))")
  echo "rep=$rep W=$w lib=$v $r"
done; done; done
