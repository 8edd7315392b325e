mkdir -p gpurun_out/r05a
BSR_ASM_STATS=1 python bench.py --cpu-sample 0 --extras 0 --steps 100 --min-time 0 2>&1 >/dev/null | grep "tile asm" | tail -2
BSR_TILE_STAMPS=1 python tools/tile_stamps.py --workload c2 --batch 64 > gpurun_out/r05a/stamps_asm.txt 2>&1
BSR_TILE_ASM=0 BSR_TILE_STAMPS=1 python tools/tile_stamps.py --workload c2 --batch 64 > gpurun_out/r05a/stamps_noasm.txt 2>&1
BSR_TILE_SPLIT=0 BSR_TILE_STAMPS=1 python tools/tile_stamps.py --workload c2 --batch 64 > gpurun_out/r05a/stamps_nosplit.txt 2>&1
head -30 gpurun_out/r05a/stamps_asm.txt; echo ----; head -30 gpurun_out/r05a/stamps_noasm.txt; echo ---; head -16 gpurun_out/r05a/stamps_nosplit.txt
