fail=0
for i in $(seq 1 8); do r=$(python -m pytest tests -m gpu -q -x 2>&1 | grep -E 'passed|failed|error' | tail -1); echo "run $i: $r"; case "$r" in *failed*|*error*|"") fail=1;; esac; done
ref=""
for i in $(seq 1 12); do o=$(python tools/chain_throughput.py --K 3 --props 20000 --chains 8 --batch 32 | sed 's/ in [0-9.]* s = [0-9]* proposals\/s//; s/init [0-9.]* s//'); if [ -z "$ref" ]; then ref="$o"; echo "$o"; fi; if [ "$o" != "$ref" ]; then echo "MISMATCH: $o"; fail=1; fi; done
for i in $(seq 1 6); do o=$(python tools/chain_throughput.py --K 5 --props 10000 --chains 16 --batch 16 | sed 's/ in [0-9.]* s = [0-9]* proposals\/s//; s/init [0-9.]* s//'); if [ $i = 1 ]; then ref2="$o"; echo "$o"; fi; if [ "$o" != "$ref2" ]; then echo "MISMATCH: $o"; fail=1; fi; done
echo "soak fail=$fail"
