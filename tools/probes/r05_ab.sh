mkdir -p gpurun_out/r05a
python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05a/bench_asm.json 2> gpurun_out/r05a/bench_asm.err
BSR_TILE_ASM=0 python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05a/bench_noasm.json 2> gpurun_out/r05a/bench_noasm.err
BSR_TILE_SPLIT=0 python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05a/bench_nosplit.json 2> gpurun_out/r05a/bench_nosplit.err
python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05a/bench_asm2.json 2> gpurun_out/r05a/bench_asm2.err
for f in asm noasm nosplit asm2; do python - <<PY
import json
d=json.loads(open("gpurun_out/r05a/bench_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d.get("roofline"))
PY
done
bash tools/profile_bench.sh r05a_c2 --workload c2 --batch 64 > gpurun_out/r05a/profile_c2.txt 2>&1
tail -12 gpurun_out/r05a/profile_c2.txt
bash tools/pmc_tile.sh --workload c2 --batch 64 > gpurun_out/r05a/pmc_c2.txt 2>&1
cp gpurun_out/pmc_tile.json gpurun_out/r05a/pmc_tile_c2_B64.json
cat gpurun_out/r05a/pmc_tile_c2_B64.json
