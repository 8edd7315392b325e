for v in "BSR_LAZY_TAIL=1 BSR_DONE_WORD=1" "BSR_LAZY_TAIL=0 BSR_DONE_WORD=1" "BSR_LAZY_TAIL=0 BSR_DONE_WORD=0"; do
  echo "== $v"
  env $v timeout 600 python -m pytest tests/test_gpu_config4.py -x -q -m gpu -k memo 2>&1 | grep "passed\|failed\|AssertionError" | tail -3
done
BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep -A4 "host cost" | tail -5
