#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# leaf batch threshold of the batched rounds, now a per-frame value (DevFrame::leaf_min)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_leafmin; mkdir -p $OUT
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
run c2_default c2
for l in 1 2 4 6 8 12 16 24; do run c2_leafmin$l c2 PBRT_HIP_LEAF_MIN=$l; done
for et in 4 12 16; do run c2_leafmin8_et$et c2 PBRT_HIP_LEAF_MIN=8 PBRT_HIP_EXIT_THRESH=$et; done
for wl in c3 p1000000; do run ${wl}_default $wl; for l in 12 16 32; do run ${wl}_leafmin$l $wl PBRT_HIP_LEAF_MIN=$l; done; done
