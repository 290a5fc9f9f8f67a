#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_overlap; mkdir -p $OUT
timeout 900 python tools/r03_vertex_check.py 2>&1 | grep -v "film_equal True counters_equal True" | tail -20 | tee $OUT/check.log
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload ${WL:-p1000000} > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame trace_ms", r["kernel_ms"], "shade_ms", r["frame_kernels_ms"]["shade_launches"], "render_ms", r["frame_kernels_ms"]["render"], "iters", r.get("pipeline_iterations"), "slots", r.get("pipeline_slots"))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
run p1m_serial PBRT_HIP_OVERLAP=0
for tc in 160 192 208 224; do run p1m_ov_tc$tc PBRT_HIP_TRACE_CUS=$tc; done
run p1m_ov_tc192_16M PBRT_HIP_TRACE_CUS=192 PBRT_HIP_PIPE_SLOTS=16777216
WL=c4 run c4_ov
WL=c5 run c5_serial PBRT_HIP_OVERLAP=0
for tc in 128 160 192; do WL=c5 run c5_ov_tc$tc PBRT_HIP_TRACE_CUS=$tc; done
