#!/bin/bash
# round 4 EXPERIMENT: one 16-byte block per two tree levels (RT_QUAD_BLOCKS variant, tools/build_variant.py quad -DRT_QUAD_BLOCKS): parity first, then timings
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe12; mkdir -p $OUT
Q=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_quad.so
PBRT_HIP_LIB_PATH=$Q timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "trace or golden or flavour or 1m_direct or 1m_path or larger" 2>&1 | tail -12 | tee $OUT/tests.txt
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
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
for wl in c3 p1000000 c4 c5; do
  run ${wl}_default $wl PBRT_HIP_TUNE=1
  run ${wl}_quad $wl PBRT_HIP_LIB_PATH=$Q
done
} 2>&1 | tee $OUT/scan.txt
