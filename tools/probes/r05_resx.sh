for cfg in "BSR_TILE_SPLIT=1" "BSR_TILE_SPLIT=3" "BSR_TILE_SPLIT=1" "BSR_TILE_SPLIT=3"; do
echo "== $cfg (3: timing only, wrong results)"
env $cfg timeout 300 python tools/probes/two_callers.py --rows 0 --callers 3 --depth 8 2>&1 | grep "caller(s)"
done
BSR_TILE_SPLIT=3 BSR_TILE_STAMPS=1 python tools/tile_stamps.py 2>&1 | grep "wave lifetime\|stage first\|all chunks"
