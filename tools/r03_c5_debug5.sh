export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
p = e.load_package(); p.build(defines=("-DRT_DEBUG_PIXEL",))
PY
for o in mega pipe; do echo "== $o"; PBRT_HIP_DEBUG_PIXEL=495,196 timeout 300 python tools/r03_c5_debug4.py $o 2>&1 | grep -v "^DEV" | grep "VOL step" | head -4; done
