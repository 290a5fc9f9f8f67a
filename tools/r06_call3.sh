#!/bin/bash
# GPU box, round 6 call 3: the whole GPU suite, the full bench line (new sub-records + parity leg), finer kd parameter scan, 10 M-triangle scene create log.
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_call3; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 1700 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 1500 $O/bench_full.json; tail -5 $O/bench_full.err
SETS="4:1:0:2 4:1:0:3 4:1:0:4 2:1:0:4 8:1:0:4 4:1:0.1:4 4:2:0:4 8:2:0:4 3:1:0:3 6:1:0:6 4:1:0:6 4:1:0:8 8:1:0:8 16:1:0:4 16:1:0:8" tools/kd_param_scan.sh r06_kd_scan2 "p1000000 c3"
