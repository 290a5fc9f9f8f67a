#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# one GPU: chunks of the work list handed out in a strided order (PBRT_HIP_SCATTER = multiplier, < 1: fraction of the chunk count)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_order; mkdir -p $OUT
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
  run sc0_$wl $wl
  for f in 0.618 0.25 0.03125 0.001 3 17 257; do run sc${f}_$wl $wl PBRT_HIP_SCATTER=$f; done
done 2>&1 | tee $OUT/scatter.txt
