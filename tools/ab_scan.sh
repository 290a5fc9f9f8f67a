#!/bin/bash
# GPU box: A/B scan of library variants (tools/build_variant.py NAME ...) over bench.py workloads, one gpurun call.
#   tools/ab_scan.sh TAG "VARIANTS" "WORKLOADS" [PYTEST_K]
# VARIANTS: names of pbrt-v1_amd/lib/libpbrt_hip_<name>.so ("default" = the product library); NAME:ENV=V,ENV2=V2 adds knobs to a run.
# WORKLOADS: bench.py --workload names (c2 c3 p1000000 c4 c5 ...).  PYTEST_K: if given, each variant first runs the parity subset
# `pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -k PYTEST_K`.  Output: gpurun_out/<TAG>/scan.txt (+ one JSON per run).
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
TAG=$1; VARIANTS=$2; WORKLOADS=$3; KEXPR=${4:-}
STEPS=${STEPS:-6}; WARMUP=${WARMUP:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib
venv() {  # variant spec -> env assignments
  local spec=$1 name=${1%%:*} extra=""
  [ "$spec" != "$name" ] && extra=$(echo "${spec#*:}" | tr ',' ' ')
  if [ "$name" = default ]; then echo "PBRT_HIP_TUNE=1 $extra"; else echo "PBRT_HIP_LIB_PATH=$L/libpbrt_hip_$name.so $extra"; fi
}
{
if [ -n "$KEXPR" ]; then
  for v in $VARIANTS; do
    echo "== parity subset, $v"
    env $(venv $v) timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "$KEXPR" 2>&1 | tail -2
  done
fi
for wl in $WORKLOADS; do
  for v in $VARIANTS; do
    tag=${wl}_$(echo $v | tr ':=,' '___')
    env $(venv $v) timeout 900 python bench.py --no-cpu-baseline --no-extra --steps $STEPS --warmup $WARMUP --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("%-40s %9.1f Mrays/s %8.3f ms/frame  kernel %8.3f ms  frac %.3f" % ("$tag", j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"]))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-800:])
PY
  done
done
} 2>&1 | tee $OUT/scan.txt
