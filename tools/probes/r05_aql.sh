export BSR_AQL_VERBOSE=1
timeout 300 python - <<'PY'
import sys, os
sys.path.insert(0, "mcmc-symreg_amd"); sys.path.insert(0, "tests")
import numpy as np
from bsr.device import DeviceContext
from bsr.node import Node
from bsr.tape import flatten
def leaf(f):
    n = Node(1); n.type = 0; n.feature = np.array([f]); return n
def un(op, c):
    n = Node(0); n.type, n.operator, n.left = 1, op, c; c.parent = n; return n
def bi(op, l, r):
    n = Node(0); n.type, n.operator, n.left, n.right = 2, op, l, r; l.parent = r.parent = n; return n
rs = np.random.RandomState(0)
N, d, K = 100000, 10, 3
X = rs.uniform(-3, 3, size=(N, d)); y = X[:, 0] * X[:, 1] + 0.1 * rs.standard_normal(N)
res = {}
for aql in ("1", "0"):
    os.environ["BSR_AQL"] = aql
    ctx = DeviceContext(X, y, K=K, n_chains=1, max_batch=64)
    for k in range(K):
        ctx.set_current(0, k, flatten([bi('*', leaf(0), leaf(1)), un('sin', leaf(2)), un('cos', leaf(4))][k]))
    ctx.refresh(0)
    B = 64
    tapes = [flatten(bi('+', un('sin', leaf(i % d)), leaf((i + 1) % d))) for i in range(B)]
    r = ctx.score_batch(tapes, np.zeros(B, np.int32), np.arange(B, dtype=np.int32) % K, np.full(B, 0.8))
    r2 = ctx.score_batch(tapes, np.zeros(B, np.int32), np.arange(B, dtype=np.int32) % K, np.full(B, 0.8))
    print("BSR_AQL", aql, ctx.dispatch_info(), r["loglik"][:3], (r.tobytes() == r2.tobytes()), flush=True)
    res[aql] = r.tobytes()
    ctx.close()
print("same bytes both ways:", res["1"] == res["0"])
PY
echo "== tests"
timeout 1500 python -m pytest tests/test_gpu_ctx_sequence.py tests/test_gpu_kernels.py tests/test_gpu_tile_asm.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|aql" | tail -5
mkdir -p gpurun_out/r05g
for v in a1 a0 a1b a0b; do
  case $v in a0*) export BSR_AQL=0;; *) unset BSR_AQL;; esac
  timeout 600 python bench.py --cpu-sample 0 --extras 0 > gpurun_out/r05g/bench_$v.json 2>gpurun_out/r05g/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05g/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", round(d["value"]), round(d["ms_per_step"]*1000,2), d["regions"]["spread"])
except Exception as e:
    print("$v failed", e); print(open("gpurun_out/r05g/bench_$v.err").read()[-800:])
PY
done
unset BSR_AQL
for v in 1 0; do BSR_AQL=$v timeout 600 python bench.py --cpu-sample 0 --extras 0 --rows 2048 > gpurun_out/r05g/rows_$v.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r05g/rows_$v.json").read().strip().splitlines()[-1])
print("rows2048 aql=$v", round(d["value"]), round(d["ms_per_step"]*1000,2))
PY
done
BSR_HOST_PROF=1 timeout 600 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep -A3 "host cost" | tail -4
