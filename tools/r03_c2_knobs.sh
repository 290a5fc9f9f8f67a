#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# C2 (tiny cache-resident tree): the host-side choices of make_frame re-scanned on the round-3 build
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_c2knobs; mkdir -p $OUT
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload c2 > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame render_ms", r["frame_kernels_ms"]["render"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-800:])
PY
}
run base
for tm in 1 2 3 4; do for ps in 0 1; do for et in 0 8 16 32; do
  run tm${tm}_ps${ps}_et${et} PBRT_HIP_TRAV_MODE=$tm PBRT_HIP_PHASE_SYNC=$ps PBRT_HIP_EXIT_THRESH=$et
done; done; done 2>&1 | tee $OUT/scan.txt
for tm in 2 3; do for et in 8 32; do
  run occ1_tm${tm}_et${et} PBRT_HIP_HIGH_OCC=1 PBRT_HIP_TRAV_MODE=$tm PBRT_HIP_EXIT_THRESH=$et
done; done 2>&1 | tee -a $OUT/scan.txt
