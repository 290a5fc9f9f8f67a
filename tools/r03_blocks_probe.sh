#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_blocks; mkdir -p $OUT
timeout 600 python tools/r03_vertex_check.py 2>&1 | grep -v "True counters_equal True" | tail -30 | tee $OUT/check.log
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
run pipe_blocks1 PBRT_HIP_PIPELINE=1 PBRT_HIP_TREELET_LOG=1
grep TREELET $OUT/pipe_blocks1.err | head -2
run pipe_blocks0 PBRT_HIP_PIPELINE=1 PBRT_HIP_PAIR_BLOCKS=0
run mega_blocks1 PBRT_HIP_PIPELINE=0
run mega_blocks0 PBRT_HIP_PIPELINE=0 PBRT_HIP_PAIR_BLOCKS=0
WL=c3 run c3_mega_blocks1 PBRT_HIP_PIPELINE=0
WL=c3 run c3_mega_blocks0 PBRT_HIP_PIPELINE=0 PBRT_HIP_PAIR_BLOCKS=0
WL=c5 run c5_blocks1
WL=c5 run c5_blocks0 PBRT_HIP_PAIR_BLOCKS=0
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "1m or million or trace" 2>&1 | tail -5
