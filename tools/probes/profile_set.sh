# The round's evidence set (GPU box): per-kernel stats, HBM traffic, instruction mix, wave stamps, the default bench line
# and the N-rank lines on the one device.  usage: bash tools/probes/profile_set.sh r03a
tag=${1:-r03a}
bash tools/profile_bench.sh ${tag}_c2 --extras 0 > gpurun_out/prof_${tag}_c2.log 2>&1
bash tools/profile_bench.sh ${tag}_c3 --extras 0 --workload c3 > gpurun_out/prof_${tag}_c3.log 2>&1
bash tools/profile_bench.sh ${tag}_c5 --extras 0 --workload c5 > gpurun_out/prof_${tag}_c5.log 2>&1
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
bash tools/pmc_tile.sh > gpurun_out/${tag}_pmc_tile.log 2>&1
cp gpurun_out/pmc_tile.json gpurun_out/${tag}_pmc_tile_c2_B64.json
BSR_TILE_STAMPS=1 python tools/tile_stamps.py > gpurun_out/${tag}_wave_stamps.txt 2>&1
# config 5 (the streaming kernel): its counters, wave stamps, per-shape executed costs; the fp32-vs-fp64 chain sweep
bash tools/pmc_tile.sh --workload c5 > gpurun_out/${tag}_pmc_tile_c5.log 2>&1
cp gpurun_out/pmc_tile.json gpurun_out/${tag}_pmc_tile_c5_B64.json
bash tools/pmc_tile.sh --workload c3 > gpurun_out/${tag}_pmc_tile_c3.log 2>&1
cp gpurun_out/pmc_tile.json gpurun_out/${tag}_pmc_tile_c3_B64.json
BSR_TILE_STAMPS=1 python tools/tile_stamps.py --workload c5 > gpurun_out/${tag}_wave_stamps_c5.txt 2>&1
bash tools/probes/op_costs.sh --N 1000000 --d 10 > gpurun_out/${tag}_op_costs_stream_d10.txt 2>&1
bash tools/probes/op_costs.sh > gpurun_out/${tag}_op_costs_c2.txt 2>&1
python tools/fp32_chain_sweep.py --out gpurun_out/${tag}_fp32_chain_sweep.json > gpurun_out/${tag}_fp32_chain_sweep.log 2>&1
./tools/micro/fp64_issue.bin > gpurun_out/${tag}_micro_fp64_issue.txt 2>&1
./tools/micro/issue_mix.bin > gpurun_out/${tag}_micro_issue_mix.txt 2>&1
BSR_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 50 --warmup 5 --cpu-sample 0 --extras 0 > gpurun_out/${tag}_bench_2ranks.json 2>/dev/null
BSR_SHARE_DEVICE=1 python bench.py --gpus 8 --steps 50 --warmup 5 --cpu-sample 0 --extras 0 > gpurun_out/${tag}_bench_8ranks.json 2>/dev/null
python tools/host_profile.py > gpurun_out/${tag}_host_profile.txt 2>&1
for w in c2 c3 c5; do
  cp gpurun_out/prof_${tag}_$w/kernel_stats.csv gpurun_out/${tag}_kernel_stats_bench_${w}_B64.csv
  cp gpurun_out/prof_${tag}_$w/traffic.json gpurun_out/${tag}_traffic_${w}_B64.json
  cp gpurun_out/prof_${tag}_$w/bench_stats.json gpurun_out/${tag}_${w}_B64_bench_under_rocprof.json
  rm -rf gpurun_out/prof_${tag}_$w/stats gpurun_out/prof_${tag}_$w/fetch gpurun_out/prof_${tag}_$w/write
done
rm -rf gpurun_out/pt1 gpurun_out/pt2 gpurun_out/pt3 gpurun_out/pt4 gpurun_out/oc1 gpurun_out/oc2
tail -c 600 gpurun_out/${tag}_bench_default.json
