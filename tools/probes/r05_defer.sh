mkdir -p gpurun_out/r05a
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --cpu-sample 0 --extras 0 $EXTRA > gpurun_out/r05a/bench_$name.json 2> gpurun_out/r05a/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05a/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"]), round(d["ms_per_step"]*1000,2))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/r05a/bench_$name.err").read()[-1500:])
PY
}
EXTRA="--rows 2048"; run rows_defer0 BSR_DEFER_STAGE=0; run rows_defer1_t2 BSR_DEFER_STAGE=1; run rows_defer1_t3 BSR_SUBMIT_THREADS=3; run rows_defer1_t4 BSR_SUBMIT_THREADS=4
EXTRA=""; run full_defer0 BSR_DEFER_STAGE=0; run full_defer1_t2 BSR_DEFER_STAGE=1; run full_defer1_t3 BSR_SUBMIT_THREADS=3; run full_defer1_t4 BSR_SUBMIT_THREADS=4
EXTRA="--depth 8"; run d8_defer1_t3 BSR_SUBMIT_THREADS=3; run d8_defer1_t4 BSR_SUBMIT_THREADS=4; run d8_t3_noasm BSR_SUBMIT_THREADS=3 BSR_TILE_ASM=0
BSR_SUBMIT_THREADS=3 BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep "host cost" | tail -1
