#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 3000 python -m pytest tests -q -m gpu -x --durations=12 "$@" 2>&1 | tail -40 | tee gpurun_out/r03/gputests.log
