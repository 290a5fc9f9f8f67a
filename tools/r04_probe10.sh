#!/bin/bash
# round 4: the scheduling constants re-scanned on the final build (XCD bands + 32x32 work blocks changed what a wave's lanes share)
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe10; mkdir -p $OUT
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 8 --warmup 2 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("%-30s %9.1f Mrays/s %8.3f ms/frame  kernel %8.3f ms  frac %.3f" % ("$tag", j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"]))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-800:])
PY
}
{
for wl in c3 p1000000; do
  run ${wl}_default $wl PBRT_HIP_TUNE=1
  for v in 16 24 40 48; do run ${wl}_exit$v $wl PBRT_HIP_EXIT_THRESH=$v; done
  for v in 12 16 32; do run ${wl}_leafmin$v $wl PBRT_HIP_LEAF_MIN=$v; done
  run ${wl}_default_again $wl PBRT_HIP_TUNE=1
done
} 2>&1 | tee $OUT/scan.txt
