#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# one GPU: the order of the work list (scanline, 2-D tiles of several shapes)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_order; mkdir -p $OUT
run() {  # tag, workload, args...
  tag=$1; wl=$2; shift; shift
  timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 --workload $wl "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame render_ms", r["frame_kernels_ms"]["render"], "gather", r["frame_kernels_ms"]["film_gather"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
for wl in c3 c2 p1000000 c4; do
  run scan_$wl $wl --tile-2d 0
  for t in 8 16 32 64 128 256; do run t${t}_$wl $wl --tile-2d $t; done
  run t256x8_$wl $wl --tile-2d 256 --tile-h 8
  run t64x4_$wl $wl --tile-2d 64 --tile-h 4
  run t16x64_$wl $wl --tile-2d 16 --tile-h 64
done 2>&1 | tee $OUT/scan.txt
