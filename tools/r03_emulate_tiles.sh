#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# rank 0's share of an 8-rank frame against the 2-D tile size (one GPU)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_emulate; mkdir -p $OUT
run() {  # tag, workload, args...
  tag=$1; wl=$2; shift; shift
  PBRT_BENCH_EMULATE_WORLD=8 timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 --workload $wl "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["ms_per_step"], "ms/frame render_ms", r["frame_kernels_ms"]["render"], "gather", r["frame_kernels_ms"]["film_gather"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
for wl in c3 p1000000; do
  for t in 16 32 64 128 256; do run n8_tile${t}_$wl $wl --tile-2d $t; done
  run n8_tile1d4096_$wl $wl --tile-2d 0 --tile-pixels 4096
done 2>&1 | tee $OUT/tiles_scan.txt
