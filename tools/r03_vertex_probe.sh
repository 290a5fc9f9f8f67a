#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_vertex; mkdir -p $OUT
timeout 600 python tools/r03_vertex_check.py 2>&1 | tail -40 | tee $OUT/check.log
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload ${WL:-p1000000} > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame trace_ms", r["kernel_ms"], "render_ms", r["frame_kernels_ms"]["render"], "iters", r.get("pipeline_iterations"), "slots", r.get("pipeline_slots"))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
run pipe_ray PBRT_HIP_PIPELINE=1 PBRT_HIP_PIPE_VERTEX=0
run pipe_vertex_8M PBRT_HIP_PIPELINE=1
run pipe_vertex_4M PBRT_HIP_PIPELINE=1 PBRT_HIP_PIPE_SLOTS=4194304
run pipe_vertex_17M PBRT_HIP_PIPELINE=1 PBRT_HIP_PIPE_SLOTS=17825792
PBRT_HIP_PIPELINE=1 PBRT_HIP_PIPE_TRACE_LOG=1 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 --workload p1000000 2>&1 | grep "^PIPE" | tail -40 > $OUT/iters_vertex.log
cat $OUT/iters_vertex.log
WL=c4 run c4_mega PBRT_HIP_PIPELINE=0
WL=c4 run c4_pipe_vertex PBRT_HIP_PIPELINE=1
WL=c4 run c4_pipe_vertex_17M PBRT_HIP_PIPELINE=1 PBRT_HIP_PIPE_SLOTS=17825792
