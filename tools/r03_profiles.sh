#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# GPU box: the round's evidence: default bench line (driver-equivalent), rocprofv3 kernel stats + PMC passes per workload, the 10 M-triangle C4 record
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( time timeout 900 python bench.py > gpurun_out/r03/bench_full.json 2> gpurun_out/r03/bench_full.err ) 2> gpurun_out/r03/bench_full.time
tail -c 600 gpurun_out/r03/bench_full.json; cat gpurun_out/r03/bench_full.time
for wl in "$@"; do
  bash tools/profile_r03.sh $wl > gpurun_out/prof_r03_$wl.log 2>&1
  python tools/publish_profile_r03.py $wl --publish > /dev/null 2>&1
  python - <<PY
import json
try:
    j = json.load(open("profiles/r03_${wl}_render_kernel.json")); print("$wl", j["kernel"], j["ms_per_frame_kernel_trace"], j["derived"])
except Exception as e: print("$wl publish failed", e)
PY
done
mkdir -p gpurun_out/r03/profiles; cp profiles/r03_* gpurun_out/r03/profiles/ 2>/dev/null
