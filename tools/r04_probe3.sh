#!/bin/bash
# round 4, third GPU call: the march kernel of the queue pipeline (C5): parity tests first, then timings of its build variants and pool sizes
export PBRT_HIP_TUNE=1
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe3c; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "c4_c5 or vol or pipeline_workloads or flavour or pool" 2>&1 | tail -15 | tee $OUT/tests.txt
L=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib
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
run c5_march5 c5 PBRT_HIP_LIB_PATH=$L/libpbrt_hip_march5.so
run c5_march6 c5 PBRT_HIP_LIB_PATH=$L/libpbrt_hip_march6.so
run c5_march5_16m c5 PBRT_HIP_LIB_PATH=$L/libpbrt_hip_march5.so PBRT_HIP_PIPE_SLOTS=16777216
} 2>&1 | tee $OUT/scan.txt
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q -k "eight or prebuilt" 2>&1 | tail -8 | tee $OUT/tests_multirank.txt
PBRT_HIP_VERIFY_DERIVED=1 timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "1m_trace or 1m_tree" 2>&1 | tail -5 | tee $OUT/tests_derived.txt
PBRT_HIP_CREATE_LOG=1 timeout 900 python bench.py --workload c4_10000000 --no-cpu-baseline --no-extra --steps 2 --warmup 1 2>&1 | grep -E "CREATE|value" | cut -c1-300 | tee $OUT/create10m.txt
