#!/bin/bash
# round 4: the device compiler's scheduling strategies on the traversal kernels (variants: tools/build_variant.py NAME --mllvm OPT --units rt_mega_d,rt_mega_p,rt_march,rt_trace)
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe13; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib
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
for v in ilp memclause relaxocc; do
  PBRT_HIP_LIB_PATH=$L/libpbrt_hip_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "timed_kernels or trace_matches" 2>&1 | tail -2
done
for wl in c3 p1000000 c5; do
  run ${wl}_default $wl PBRT_HIP_TUNE=1
  for v in ilp memclause relaxocc; do run ${wl}_$v $wl PBRT_HIP_LIB_PATH=$L/libpbrt_hip_$v.so; done
done
} 2>&1 | tee $OUT/scan.txt
