#!/bin/bash
# GPU box, round 6 call 2: the whole GPU suite on the entry layout, L2 / L1 counters for the two layouts, the runs form on tiny scenes, kd parameter scan.
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_call2; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
STEPS=6 WARMUP=2 tools/ab_scan.sh r06_runs_scan "default default:PBRT_HIP_LEAF_RUNS=1 default:PBRT_HIP_LEAF_RUNS=0 r05" "c2 p1000000 c3"
tools/pmc_probe.sh r06_pmc_p1m_entries p1000000
tools/pmc_probe.sh r06_pmc_p1m_r05 p1000000 r05
tools/kd_param_scan.sh r06_kd_scan "p1000000 c3"
