#!/bin/bash
# full default bench (all legs) per BSR_AUX_CUS value; JSON lines under gpurun_out/aux_full/
mkdir -p gpurun_out/aux_full
for v in ${AUX_LIST:-0 48 64 80}; do
  BSR_AUX_CUS=$v python bench.py --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/aux_full/aux_$v.json
  python3 - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open(f"gpurun_out/aux_full/aux_{v}.json"))
ex = d.get("extra", {})
def g(k, *path):
    x = ex.get(k, {})
    for p in path:
        x = x.get(p, {}) if isinstance(x, dict) else {}
    return x
print("AUX", v, "c2 %.3f M/s %.2f us" % (d["value"] / 1e6, d["ms_per_step"] * 1e3),
      "| c3", g("c3", "value"), "| c5", g("c5", "value"), "| c4", g("c4", "value"),
      "| engine", {k: ex.get("c4_native_engine", {}).get(k) for k in ("consumed_per_s", "scored_per_s", "discarded_share")})
PY
done
