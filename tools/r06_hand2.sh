#!/bin/bash
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
STEPS=6 WARMUP=2 tools/ab_scan.sh r06_hand_scan2 "hand2 hand8 hand:PBRT_HIP_EXIT_THRESH=24 hand:PBRT_HIP_EXIT_THRESH=48 hand2:PBRT_HIP_EXIT_THRESH=40 default" "p1000000 c2"
for v in hand2; do
  for n in 8; do
    for lib in $v ""; do
      [ -n "$lib" ] && export PBRT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_$lib.so || unset PBRT_HIP_LIB_PATH
      for wl in p1000000 c4; do
      PBRT_BENCH_EMULATE_WORLD=$n python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 1 --workload $wl 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl N=$n lib=${lib:-default} step %.3f ms kernel %.3f ms' % (j['ms_per_step'], j['roofline']['kernel_ms']))" | tee -a gpurun_out/r06_hand_scan2/scan.txt
      done
    done
  done
done
