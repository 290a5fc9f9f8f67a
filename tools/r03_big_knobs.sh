#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# the 1 M-triangle frames: host-side choices re-scanned on the final round-3 build
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_bigknobs; mkdir -p $OUT
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
for wl in c3 p1000000; do
  run ${wl}_base $wl
  for tm in 1 2 4; do for ps in 0 1; do for et in 24 32 40; do
    run ${wl}_tm${tm}_ps${ps}_et${et} $wl PBRT_HIP_TRAV_MODE=$tm PBRT_HIP_PHASE_SYNC=$ps PBRT_HIP_EXIT_THRESH=$et
  done; done; done
  run ${wl}_occ0 $wl PBRT_HIP_HIGH_OCC=0
done 2>&1 | tee $OUT/scan.txt
