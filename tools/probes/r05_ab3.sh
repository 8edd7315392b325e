mkdir -p gpurun_out/r05a
for v in 1 0 2; do
  BSR_TILE_ASM=$v BSR_TILE_STAMPS=1 python tools/tile_stamps.py --workload c2 --batch 64 > gpurun_out/r05a/stamps_mode$v.txt 2>&1
  echo "== BSR_TILE_ASM=$v"; sed -n '1p;5p;7p;9,12p;14p' gpurun_out/r05a/stamps_mode$v.txt
done
BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep -i "host\|stage\|us" | tail -8
python tools/host_profile.py 2>&1 | tail -5
