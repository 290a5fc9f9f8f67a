#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# film gather strip height against the number of shards (rank 0's share on one GPU)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_emulate; mkdir -p $OUT
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["ms_per_step"], "ms/frame render_ms", r["frame_kernels_ms"]["render"], "gather", r["frame_kernels_ms"]["film_gather"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
for wl in c3 c2; do
  for n in 2 4 8; do
    for r in 8 16 32 64; do run n${n}_rows${r}_$wl $wl PBRT_BENCH_EMULATE_WORLD=$n PBRT_HIP_GATHER_ROWS=$r; done
  done
done 2>&1 | tee $OUT/rows_scan.txt
