export BSR_AQL_ROW_QUEUES=0
for rows in 0 2048; do
for cfg in "BSR_AQL=0" "BSR_AQL=1" "BSR_AQL_TEST_NOBAR=1" "BSR_AQL_TEST_NOBAR=2" "BSR_AQL_TEST_NOBAR=2 BSR_SUBMIT_THREADS=3"; do
echo "== rows $rows $cfg (NOBAR: timing only, results wrong)"
env $cfg timeout 300 python tools/probes/two_callers.py --rows $rows --callers 1 --depth 8 2>&1 | grep "caller(s)"
done
done
