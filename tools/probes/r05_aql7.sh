bash tools/timeline.sh --depth 8 2>&1 | head -12
cp gpurun_out/tl_bench.json gpurun_out/r05g/tl_bench_aql.json 2>/dev/null
BSR_AQL=0 bash tools/timeline.sh 2>&1 | head -8
