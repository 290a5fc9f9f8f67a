#!/bin/bash
# round 4, fifth GPU call: the list-driven march-step kernel (C5)
export PBRT_HIP_TUNE=1
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe5c; mkdir -p $OUT
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "c4_c5 or vol or pipeline_workloads or flavour or pool" 2>&1 | tail -12 | tee $OUT/tests.txt
grep -q " passed" $OUT/tests.txt && ! grep -q "failed" $OUT/tests.txt || { echo "parity tests failed: no timings"; exit 1; }
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame", r["kernel"][-60:], "render_ms", r["frame_kernels_ms"]["render"], "shade", r["frame_kernels_ms"]["shade_launches"], "gather", r["frame_kernels_ms"]["film_gather"], "iters", r.get("pipeline_iterations"), "slots", r.get("pipeline_slots"), "frac", r["frac"], "frac_frame", r["frac_frame_kernels"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
{
run c5_default c5
run c5_marchnat c5 PBRT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_marchnat.so
run c5_slots16m c5 PBRT_HIP_PIPE_SLOTS=16777216
run c5_slots24m c5 PBRT_HIP_PIPE_SLOTS=25165824
} 2>&1 | tee $OUT/scan.txt
timeout 900 python -m pytest tests/test_parity_chain.py -m gpu -x -q -k "api_" 2>&1 | tail -4 | tee $OUT/tests_api.txt
