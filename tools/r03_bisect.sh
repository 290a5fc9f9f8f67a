#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
cd $GRAFT_REPO_ROOT
for lvl in 7 11; do
python - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
p = e.load_package(); p.build(defines=("-DRT_PIPE_VOL_SLIM=$lvl",))
PY
echo "--- SLIM=$lvl"
python tools/r03_mix6.py 6 2>&1 | grep "equal" | cut -c1-90
done
