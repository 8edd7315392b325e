cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in c3; do for B in 64 256; do out=gpurun_out/prof_${wl}_B$B; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 50 --warmup 5 --cpu-sample 0 --workload $wl --batch $B > $out/bench.json 2> $out/err
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv; rm -rf $out/stats
head -8 $out/kernel_stats.csv | cut -c1-200; cat $out/bench.json | cut -c1-400; done; done
