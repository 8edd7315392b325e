for t in 1 2 3; do
echo "== submit threads $t, defer 0, rows 2048"; BSR_DEFER_STAGE=0 BSR_SUBMIT_THREADS=$t BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 --rows 2048 2>&1 >/dev/null | grep -A1 "host cost" | tail -2
done
echo "== full N, threads 2, defer 0"; BSR_DEFER_STAGE=0 BSR_HOST_PROF=1 python bench.py --cpu-sample 0 --extras 0 2>&1 >/dev/null | grep -A1 "host cost" | tail -2
