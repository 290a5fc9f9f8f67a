#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# rank 0's share of an N-rank frame on one GPU (PBRT_BENCH_EMULATE_WORLD): render + film gather per rank, staged vs slot gather
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_emulate; mkdir -p $OUT
true
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["ms_per_step"], "ms/frame render_ms", r["frame_kernels_ms"]["render"], "gather", r["frame_kernels_ms"]["film_gather"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
for wl in c3 c2 p1000000; do
  for n in 1 2 4 8; do
    run n${n}_slot_$wl $wl PBRT_BENCH_EMULATE_WORLD=$n
  done
  run n8_staged_$wl $wl PBRT_BENCH_EMULATE_WORLD=8 PBRT_HIP_GATHER=staged
done 2>&1 | tee $OUT/scan.txt
