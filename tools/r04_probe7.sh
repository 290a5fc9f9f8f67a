#!/bin/bash
# round 4: non-temporal loads in the traversal (1 = leaf triangle records, 2 = second pair record, 4 = first)
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe7; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame", "kernel_ms", r["kernel_ms"], "render_ms", r["frame_kernels_ms"]["render"], "frac", r["frac"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
{
for wl in p1000000 c3 c5; do
  run ${wl}_base $wl
  for v in 1 3 7; do run ${wl}_nt$v $wl PBRT_HIP_LIB_PATH=$L/libpbrt_hip_nt$v.so; done
done
} 2>&1 | tee $OUT/scan.txt
