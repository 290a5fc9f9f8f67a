#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# GPU box: does ordering the ray queue by entry point pay?  1 M-triangle path frame through the queue pipeline, unsorted vs sorted
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_sort; mkdir -p $OUT
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload ${WL:-p1000000} > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame trace_ms", r["kernel_ms"], "render_ms", r["frame_kernels_ms"]["render"], "iters", r.get("pipeline_iterations"))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
run mega PBRT_HIP_PIPELINE=0
run pipe_unsorted PBRT_HIP_PIPELINE=1
for bb in 0 6 9 12 15 18; do
  run pipe_sort_b$bb PBRT_HIP_PIPELINE=1 PBRT_HIP_SORT=1 PBRT_HIP_SORT_BEGIN_BIT=$bb
done
PBRT_HIP_PIPELINE=1 PBRT_HIP_SORT=1 PBRT_HIP_PIPE_TRACE_LOG=1 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 --workload p1000000 2>&1 | grep "^PIPE" | tail -40 > $OUT/iters_sorted.log
PBRT_HIP_PIPELINE=1 PBRT_HIP_PIPE_TRACE_LOG=1 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 --workload p1000000 2>&1 | grep "^PIPE" | tail -40 > $OUT/iters_unsorted.log
paste $OUT/iters_unsorted.log $OUT/iters_sorted.log | head -40
WL=c3 run c3_mega PBRT_HIP_PIPELINE=0
WL=c3 run c3_pipe PBRT_HIP_PIPELINE=1
WL=c3 run c3_pipe_sort PBRT_HIP_PIPELINE=1 PBRT_HIP_SORT=1
