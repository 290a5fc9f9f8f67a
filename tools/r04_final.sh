#!/bin/bash
# round 4, final GPU call: rocprofv3 evidence for the five workloads (published under profiles/), then the default bench.py run
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_final5; mkdir -p $OUT
for wl in c3 p1000000 c4 c2 c5; do
  ROUND=r04 bash tools/profile.sh $wl --publish > $OUT/profile_$wl.log 2>&1
  tail -c 600 $OUT/profile_$wl.log
done
mkdir -p $OUT/profiles; cp profiles/r04_*_kernel_stats.csv profiles/r04_*_summary.json profiles/r04_*_render_kernel.json $OUT/profiles/ 2>/dev/null
unset PBRT_HIP_TUNE
timeout 1500 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -c 1500 $OUT/bench_full.json
export PBRT_HIP_TUNE=1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/tests_all.txt
