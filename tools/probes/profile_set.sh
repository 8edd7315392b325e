bash tools/profile_bench.sh r02h_c2 --extras 0 > gpurun_out/prof_r02h_c2.log 2>&1
bash tools/profile_bench.sh r02h_c3 --extras 0 --workload c3 > gpurun_out/prof_r02h_c3.log 2>&1
bash tools/profile_bench.sh r02h_c5 --extras 0 --workload c5 > gpurun_out/prof_r02h_c5.log 2>&1
python bench.py > gpurun_out/r02h_bench_default.json 2> gpurun_out/r02h_bench_default.err
bash tools/pmc_tile.sh > gpurun_out/r02h_pmc_tile.log 2>&1
BSR_TILE_STAMPS=1 python tools/tile_stamps.py > gpurun_out/r02h_wave_stamps.txt 2>&1
BSR_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 50 --warmup 5 --cpu-sample 0 > gpurun_out/r02h_bench_2ranks.json 2>/dev/null
tail -c 600 gpurun_out/r02h_bench_default.json
