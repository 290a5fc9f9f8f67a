#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# GPU box: first contact of the queue pipeline -- parity, then the 1 M-triangle path frame with both architectures
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_pipe1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu > $OUT/test_configs.log 2>&1; tail -5 $OUT/test_configs.log
PBRT_HIP_PIPELINE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/test_parity_pipe.log 2>&1; tail -5 $OUT/test_parity_pipe.log
for wl in p1000000 c3; do
  for pipe in 0 1; do
    PBRT_HIP_PIPELINE=$pipe timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $wl > $OUT/bench_${wl}_pipe$pipe.json 2> $OUT/bench_${wl}_pipe$pipe.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/bench_${wl}_pipe$pipe.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print("$wl pipe=$pipe", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame kernel_ms", r["kernel_ms"], "frac", r["frac"], r.get("frame_kernels_ms"), r.get("pipeline_iterations"))
except Exception as e:
    print("$wl pipe=$pipe FAILED", e); print(open("$OUT/bench_${wl}_pipe$pipe.err").read()[-1500:])
PY
  done
done
