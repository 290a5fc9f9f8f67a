#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# film gather: march kernel vs the staged one (PBRT_HIP_GATHER), strip heights; parity tests first
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_gather; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_multirank_gpu.py -q -m gpu -x 2>&1 | tail -15 | tee $OUT/tests.log
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame render_ms", r["frame_kernels_ms"]["render"], "gather", r["frame_kernels_ms"]["film_gather"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
for wl in c3 c2; do
  run staged_$wl $wl PBRT_HIP_GATHER=staged
  run march_$wl $wl PBRT_HIP_GATHER=march
  run slot_$wl $wl
  for r in 8 16 32 64 128; do run slot_r${r}_$wl $wl PBRT_HIP_GATHER_ROWS=$r; done
done 2>&1 | tee $OUT/scan.txt
