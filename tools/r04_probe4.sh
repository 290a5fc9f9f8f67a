#!/bin/bash
# round 4, fourth GPU call: march kernel follow-ups (table-row prefetch, take thresholds), 5-wave megakernel with a 10-entry LDS stack, C5 profile
export PBRT_HIP_TUNE=1
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "c4_c5 or vol or pipeline_workloads" 2>&1 | tail -5 | tee $OUT/tests.txt
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
for t in 4 8 32; do run c5_take$t c5 PBRT_HIP_LIB_PATH=$L/libpbrt_hip_take$t.so; done
for wl in p1000000 c3 c4; do
  run ${wl}_default $wl
  run ${wl}_occ5s10 $wl PBRT_HIP_LIB_PATH=$L/libpbrt_hip_occ5s10.so
done
} 2>&1 | tee $OUT/scan.txt
ROUND=r04 bash tools/profile.sh c5 > $OUT/profile_c5.log 2>&1
python - <<PY
import json
j = json.load(open("$GRAFT_REPO_ROOT/gpurun_out/prof_r04_c5/summary.json"))
for n, k in j["kernels"].items(): print(n[:70], k)
for n, p in j["pmc_per_frame"].items():
    if "<false" in n: print(n[:70], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items() if k != "derived"}, p["derived"])
PY
