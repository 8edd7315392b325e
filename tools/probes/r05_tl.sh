mkdir -p gpurun_out/r05a
bash tools/timeline.sh > gpurun_out/r05a/timeline_c2_asm.txt 2>&1
BSR_TILE_ASM=0 bash tools/timeline.sh > gpurun_out/r05a/timeline_c2_noasm.txt 2>&1
head -12 gpurun_out/r05a/timeline_c2_asm.txt; echo; head -12 gpurun_out/r05a/timeline_c2_noasm.txt; sed -n '13,40p' gpurun_out/r05a/timeline_c2_asm.txt
