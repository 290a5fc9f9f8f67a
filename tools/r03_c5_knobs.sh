#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# C5 (queue pipeline): pool size re-scanned on the final build
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_c5knobs; mkdir -p $OUT
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 2 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame trace_ms", r["kernel_ms"], "shade_ms", r["frame_kernels_ms"]["shade_launches"], "iters", r.get("pipeline_iterations"), "slots", r.get("pipeline_slots"))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-800:])
PY
}
for sl in 8388608 16777216 25165824 33554432 50331648 67108864; do run c5_slots$sl c5 PBRT_HIP_PIPE_SLOTS=$sl; done
# the 1 M path frame forced through the pipeline (by vertex): does a bigger pool help there too?
for sl in 8388608 16777216; do run p1m_pipe_slots$sl p1000000 PBRT_HIP_PIPELINE=1 PBRT_HIP_PIPE_SLOTS=$sl; done
