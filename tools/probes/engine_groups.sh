#!/bin/bash
# native sampler end to end (8 chains, N=100k, d=10): chain groups in flight x proposals per chain and batch
for rep in 1 2; do for K in 3 8; do for g in ${GROUPS_LIST:-2 4 8}; do for b in ${BATCHES:-32}; do
  r=$(BSR_ENGINE_GROUPS=$g python tools/chain_throughput.py --K $K --props 20000 --chains 8 --batch $b 2>/dev/null | tail -1 | grep -o "= [0-9]* proposals/s\|discarded [0-9]*")
  echo "rep=$rep K=$K groups=$g batch=$b $r"
done; done; done; done
