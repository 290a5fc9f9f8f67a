#!/bin/bash
# GPU box, round 6 call 4: tie probe (c3_tuned's one extra reference ray), three-level blocks (RT_KD3) parity + timing
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_call4; mkdir -p $O
timeout 900 python tools/r06_tie_probe.py 2>&1 | tail -40 | tee $O/tie_probe.txt
PBRT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_kd3.so PBRT_HIP_CREATE_LOG=1 timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 2 --warmup 1 --workload p1000000 2>&1 | grep -E "CREATE|value" | cut -c1-300 | tee $O/kd3_create.txt
STEPS=6 WARMUP=2 tools/ab_scan.sh r06_kd3_scan "kd3 default" "p1000000 c3 c4 c5 c2" "test"
