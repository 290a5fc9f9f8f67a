#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_all; mkdir -p $OUT
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload ${WL:-p1000000} > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame trace_ms", r["kernel_ms"], "shade_ms", r["frame_kernels_ms"]["shade_launches"], "render_ms", r["frame_kernels_ms"]["render"], "gather", r["frame_kernels_ms"]["film_gather"], "iters", r.get("pipeline_iterations"), "frac", r["frac"], "frac_frame", r["frac_frame_kernels"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
WL=c2 run c2
WL=c3 run c3
WL=c3 run c3_occ0 PBRT_HIP_HIGH_OCC=0
WL=c3 run c3_pipe PBRT_HIP_PIPELINE=1
run p1m
run p1m_mega PBRT_HIP_PIPELINE=0
run p1m_mega_occ0 PBRT_HIP_PIPELINE=0 PBRT_HIP_HIGH_OCC=0
WL=c4 run c4
WL=c4 run c4_mega PBRT_HIP_PIPELINE=0
WL=c5 run c5
WL=c5 run c5_mega PBRT_HIP_PIPELINE=0
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
