python tools/probes/two_callers.py 2>&1 | tail -4
echo "== rows 2048"; python tools/probes/two_callers.py --rows 2048 2>&1 | tail -4
echo "== no asm"; BSR_TILE_ASM=0 python tools/probes/two_callers.py --callers 2 2>&1 | tail -3
BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep -A2 "host cost" | tail -3
