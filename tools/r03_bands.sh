#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# banded frames: equality test, then frame times against the number of bands
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_bands; mkdir -p $OUT
true
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 --workload $wl --tile-2d 0 > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame render_ms", r["frame_kernels_ms"]["render"], "gather", r["frame_kernels_ms"]["film_gather"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
for wl in c3 c2 p1000000; do
  for b in 1 2 3 4 6 8; do run bands${b}_$wl $wl PBRT_HIP_BANDS=$b; done
done 2>&1 | tee $OUT/scan.txt
