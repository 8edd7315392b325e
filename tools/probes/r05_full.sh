timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
mkdir -p gpurun_out/r05h
timeout 900 python bench.py > gpurun_out/r05h/bench_default.json 2> gpurun_out/r05h/bench_default.err
tail -c 3000 gpurun_out/r05h/bench_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
