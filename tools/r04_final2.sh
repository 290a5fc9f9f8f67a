#!/bin/bash
# round 4: the default bench.py run (device-resident step + host hand-over loop), the multi-rank tests that drive bench.py, smoke()
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_final2; mkdir -p $OUT
timeout 1500 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -c 600 $OUT/bench_full.json; tail -3 $OUT/bench_full.err
timeout 1500 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/tests_multirank.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
