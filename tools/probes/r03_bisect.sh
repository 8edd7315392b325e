for e in "X=1" "BSR_DERIVED=0" "BSR_CHAIN_EVAL=0" "BSR_REORDER=0" "BSR_TILE=0" "BSR_TILE_T=1" "BSR_TILE_T=4"; do
  echo "== $e"; env $e python bench.py --extras 0 --cpu-sample 0 --workload c5 --min-time 0.1 2>&1 | tail -1 | cut -c1-150
done
