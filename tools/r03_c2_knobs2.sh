#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# C2: around the high-occupancy flavour with batched rounds
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_c2knobs; mkdir -p $OUT
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame render_ms", r["frame_kernels_ms"]["render"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-800:])
PY
}
for tm in 1 2; do for ps in 0 1; do for et in 0 4 8 12 16 24; do
  run c2_occ1_tm${tm}_ps${ps}_et${et} c2 PBRT_HIP_HIGH_OCC=1 PBRT_HIP_TRAV_MODE=$tm PBRT_HIP_PHASE_SYNC=$ps PBRT_HIP_EXIT_THRESH=$et
done; done; done 2>&1 | tee $OUT/scan2.txt
# the other tiny-tree frame of the suite (C1-like: Cornell, whitted / direct) must not lose
for integ in c1; do run ${integ}_base $integ; run ${integ}_occ1_tm2_et8 $integ PBRT_HIP_HIGH_OCC=1 PBRT_HIP_TRAV_MODE=2 PBRT_HIP_EXIT_THRESH=8; done 2>&1 | tee -a $OUT/scan2.txt
