# Round-5 measurement set (run on the GPU box): default bench line, kernel stats + traffic (rocprofv3, separate PMC passes),
# instruction mix (PMC), wave stamps, kernel timeline for C2 / C3 / C5.  Summaries land in gpurun_out/r05n/.
mkdir -p gpurun_out/r05n
python bench.py > gpurun_out/r05n/bench_default.json 2> gpurun_out/r05n/bench_default.err
for w in c2 c3 c5; do
  bash tools/profile_bench.sh r05n_$w --workload $w --batch 64 > gpurun_out/r05n/profile_$w.txt 2>&1
  cp gpurun_out/prof_r05n_$w/kernel_stats.csv gpurun_out/r05n/kernel_stats_bench_${w}_B64.csv
  cp gpurun_out/prof_r05n_$w/traffic.json gpurun_out/r05n/traffic_${w}_B64.json
  cp gpurun_out/prof_r05n_$w/bench_stats.json gpurun_out/r05n/${w}_B64_bench_under_rocprof.json
  bash tools/pmc_tile.sh --workload $w --batch 64 > gpurun_out/r05n/pmc_$w.txt 2>&1
  cp gpurun_out/pmc_tile.json gpurun_out/r05n/pmc_tile_${w}_B64.json
done
BSR_TILE_STAMPS=1 python tools/tile_stamps.py --workload c2 --batch 64 > gpurun_out/r05n/wave_stamps_c2.txt 2>&1
bash tools/timeline.sh > gpurun_out/r05n/timeline_c2.txt 2>&1
BSR_TILE_ASM=0 bash tools/pmc_tile.sh --workload c2 --batch 64 > gpurun_out/r05n/pmc_c2_noasm.txt 2>&1
cp gpurun_out/pmc_tile.json gpurun_out/r05n/pmc_tile_c2_B64_k_tile1.json
rm -rf gpurun_out/prof_r05n_* gpurun_out/pt1 gpurun_out/pt2 gpurun_out/pt3 gpurun_out/pt4 gpurun_out/tl
tail -c 1500 gpurun_out/r05n/bench_default.json
