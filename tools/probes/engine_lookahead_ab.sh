# native sampler, 8 chains x 32, N=100k, K=3: chain groups x lookahead (GPU box)
for i in 1 2 3; do for g in 4 8; do for la in 1 0; do
echo "groups=$g lookahead=$la: $(BSR_ENGINE_GROUPS=$g BSR_ENGINE_LOOKAHEAD=$la python tools/chain_throughput.py --props 20000 2>&1 | tail -1 | sed 's/.*= \([0-9]*\) proposals.s.*discarded \([0-9]*\),.*/\1 consumed\/s, \2 discarded/')"
done; done; done
