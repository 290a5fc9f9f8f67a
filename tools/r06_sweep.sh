#!/bin/bash
# GPU box, round 6: the whole GPU suite three times -- as shipped, with the entry form forced on every scene (PBRT_HIP_LEAF_RUNS=0) and with the runs form forced on every scene (=1: also the 1 M- and 10 M-triangle trees)
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_sweep; mkdir -p $O
{ echo "== as shipped"; python -m pytest tests -m gpu -q 2>&1 | tail -3
  echo "== PBRT_HIP_LEAF_RUNS=0 (entries everywhere)"; PBRT_HIP_LEAF_RUNS=0 python -m pytest tests -m gpu -q 2>&1 | tail -3
  echo "== PBRT_HIP_LEAF_RUNS=1 (runs everywhere)"; PBRT_HIP_LEAF_RUNS=1 python -m pytest tests -m gpu -q 2>&1 | tail -3; } | tee $O/sweep.txt
