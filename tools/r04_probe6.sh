#!/bin/bash
# round 4, sixth GPU call: v_bfi component selects in the two-level step (A/B), then the whole GPU test suite
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "1m_trace or flavour or 1m_path" 2>&1 | tail -5 | tee $OUT/tests_quick.txt
grep -q " passed" $OUT/tests_quick.txt && ! grep -q "failed" $OUT/tests_quick.txt || { echo "parity tests failed: no timings"; exit 1; }
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame", r["kernel"][-50:], "render_ms", r["frame_kernels_ms"]["render"], "shade", r["frame_kernels_ms"]["shade_launches"], "gather", r["frame_kernels_ms"]["film_gather"], "iters", r.get("pipeline_iterations"), "frac", r["frac"], "frac_frame", r["frac_frame_kernels"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
{
for wl in p1000000 c3 c4 c5 c2; do
  run ${wl}_nobfi $wl PBRT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_nobfi.so
  run ${wl}_bfi $wl
done
} 2>&1 | tee $OUT/scan.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/tests_all.txt
