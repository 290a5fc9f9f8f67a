#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r03/gputests.log
