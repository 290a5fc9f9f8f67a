#!/bin/bash
# round 4, second GPU call: microbench follow-ups (scalar unit, v_cndmask forms), r03 library vs the pruned / micro-optimised one,
# C5 through the volume megakernel, the 10 M-triangle tests and the C4 frame at its stated size
export PBRT_HIP_TUNE=1
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe2; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/valu_issue_bench.hip -o /tmp/valu_issue_bench 2>/dev/null || exit 1
  timeout 300 /tmp/valu_issue_bench 20000 0 9 ) > $OUT/valu_issue2.txt 2>&1
R03=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_r03.so
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame trace_ms", r["kernel_ms"], "render_ms", r["frame_kernels_ms"]["render"], "shade", r["frame_kernels_ms"]["shade_launches"], "gather", r["frame_kernels_ms"]["film_gather"], "iters", r.get("pipeline_iterations"), "frac", r["frac"], "frac_frame", r["frac_frame_kernels"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
{
for wl in p1000000 c3 c5 c2; do
  run ${wl}_r03 $wl PBRT_HIP_LIB_PATH=$R03
  run ${wl}_new $wl
done
run p1m_pipe_r03 p1000000 PBRT_HIP_PIPELINE=1 PBRT_HIP_LIB_PATH=$R03
run p1m_pipe_new p1000000 PBRT_HIP_PIPELINE=1
run c5_mega_occ0 c5 PBRT_HIP_PIPELINE=0 PBRT_HIP_HIGH_OCC=0
run c5_mega_occ1 c5 PBRT_HIP_PIPELINE=0 PBRT_HIP_HIGH_OCC=1
} 2>&1 | tee $OUT/scan.txt
timeout 1800 python -m pytest tests/test_gpu_c4_full.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/tests_c4full.txt
PBRT_HIP_CREATE_LOG=1 timeout 1500 python bench.py --workload c4full --no-cpu-baseline --no-extra --steps 2 --warmup 1 > $OUT/c4_full.json 2> $OUT/c4_full.err
tail -c 1500 $OUT/c4_full.json; grep CREATE $OUT/c4_full.err | tail -12
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_multirank_gpu.py -m gpu -x -q -k "full_size or prebuilt or flavour or 1m_" 2>&1 | tail -15 | tee $OUT/tests_changed.txt
