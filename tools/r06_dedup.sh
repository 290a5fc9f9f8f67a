#!/bin/bash
# GPU box, round 6: one record per primitive (leaf entries) against round 5's per-reference copies -- parity first, then timing, then counters.
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_dedup; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_parity_chain.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_subset.txt
STEPS=6 WARMUP=2 tools/ab_scan.sh r06_dedup_scan "default r05 default:PBRT_HIP_LEAF_COPIES=1" "p1000000 c3 c4 c5 c2"
