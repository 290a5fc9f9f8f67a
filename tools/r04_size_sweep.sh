#!/bin/bash
# round 4: do the scheduling constants picked on the Cornell box and the 1 M-triangle soup hold in between?  
# DirectLighting (c3_N) and path (pN) frames at 10 k and 100 k soup triangles: the defaults against each knob moved on its own.
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_size_sweep; mkdir -p $OUT
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("%-34s %9.1f Mrays/s %8.3f ms/frame  kernel %8.3f ms  frac %.3f" % ("$tag", j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"]))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-800:])
PY
}
{
for wl in c3_10000 c3_100000 p10000 p100000; do
  run ${wl}_default $wl PBRT_HIP_TUNE=1
  run ${wl}_occ0 $wl PBRT_HIP_HIGH_OCC=0
  run ${wl}_exit0 $wl PBRT_HIP_EXIT_THRESH=0
  run ${wl}_exit8 $wl PBRT_HIP_EXIT_THRESH=8
  run ${wl}_exit48 $wl PBRT_HIP_EXIT_THRESH=48
  run ${wl}_phase1 $wl PBRT_HIP_PHASE_SYNC=1
  run ${wl}_leafmin8 $wl PBRT_HIP_LEAF_MIN=8
  run ${wl}_bands_flip $wl PBRT_HIP_XCD_BANDS=$([ ${wl:0:1} = c ] && echo 0 || echo 1)
done
} 2>&1 | tee $OUT/scan.txt
