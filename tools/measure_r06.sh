# Round-6 measurement set (run on the GPU box): everything DESIGN section 7 and profiles/INDEX.md cite, into gpurun_out/r06/.
# (The round's LAST builds re-ran parts of it one by one -- `python bench.py`, `bash tools/r06_stats.sh <tag> <workload> [--depth 1]`,
# `bash tools/engine_cpus.sh`, `python tools/cu_occupancy.py` -- after every change to k_solve and the sampler; profiles/INDEX.md
# names the command behind each file.  The PMC / traffic passes are from the first full run: the row passes did not change.)
#   bash tools/measure_r06.sh            (about 12 minutes)
mkdir -p gpurun_out/r06
o=gpurun_out/r06
python bench.py > $o/bench_default.json 2> $o/bench_default.err
for w in c2 c3 c5; do
  bash tools/profile_bench.sh r06_$w --workload $w --batch 64 > $o/profile_$w.txt 2>&1
  cp gpurun_out/prof_r06_$w/kernel_stats.csv $o/kernel_stats_bench_${w}_B64.csv
  cp gpurun_out/prof_r06_$w/traffic.json $o/traffic_${w}_B64.json
  cp gpurun_out/prof_r06_$w/bench_stats.json $o/${w}_B64_bench_under_rocprof.json
done
bash tools/profile_bench.sh r06_c5f32 --workload c5 --batch 64 --dtype f32 > $o/profile_c5_f32.txt 2>&1
cp gpurun_out/prof_r06_c5f32/kernel_stats.csv $o/kernel_stats_bench_c5_f32_B64.csv
cp gpurun_out/prof_r06_c5f32/traffic.json $o/traffic_c5_f32_B64.json
for w in c2 c3; do   # the pipelined default too: what the kernels take while eight (six) batches share the chip
  bash tools/r06_stats.sh r06 $w > /dev/null 2>&1
  mv $o/kernel_stats_bench_${w}.csv $o/kernel_stats_bench_${w}_B64_pipelined.csv
done
bash tools/pmc_tile.sh --workload c2 --batch 64 > $o/pmc_c2.txt 2>&1; cp gpurun_out/pmc_tile.json $o/pmc_tile_c2_B64.json
bash tools/pmc_tile.sh --workload c5 --batch 64 > $o/pmc_c5.txt 2>&1; cp gpurun_out/pmc_tile.json $o/pmc_tile_c5_B64.json
bash tools/pmc_tile.sh --workload c5 --batch 64 --dtype f32 > $o/pmc_c5_f32.txt 2>&1; cp gpurun_out/pmc_tile.json $o/pmc_tile_c5_f32_B64.json
for d in 8 4; do python tools/cu_occupancy.py --depth $d > $o/cu_occupancy_c2_depth$d.txt 2>&1; done
python tools/fp32_chain_sweep.py --out $o/fp32_chain_sweep.json > $o/fp32_chain_sweep.txt 2>&1
bash tools/engine_cpus.sh $o/engine_cpus.txt > /dev/null 2>&1
rm -rf gpurun_out/prof_r06_* gpurun_out/pt1 gpurun_out/pt2 gpurun_out/pt3 gpurun_out/pt4 $o/stats_*
tail -c 600 $o/bench_default.json
