cp build_variants/lib_stamps.so mcmc-symreg_amd/bsr/libbsr_hip.so
python tools/stamps.py 2>&1 | grep -E "kernel us|wall clock|waves with|histogram|  waves|waves stamped|lifetime|setup|sweep|reductions"
