#!/bin/bash
# GPU box: rank 0's share of an N-rank job on ONE GPU (PBRT_BENCH_EMULATE_WORLD): render of its tiles, film gather, and the merge's own kernels
# (rt_film_pack_parts, per-rank RGBA resolve; no collective), N = 1 2 4 8.   tools/emulate_world.sh TAG "WORKLOADS" [LIB_VARIANT]
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
TAG=$1; WORKLOADS=$2; VAR=${3:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
[ -n "$VAR" ] && export PBRT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_$VAR.so
{
for wl in $WORKLOADS; do
  for n in 1 2 4 8; do
    PBRT_BENCH_EMULATE_WORLD=$n timeout 1200 python bench.py --no-cpu-baseline --no-extra --steps ${STEPS:-5} --warmup 1 --workload $wl > $OUT/${wl}_n$n.json 2> $OUT/${wl}_n$n.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/${wl}_n$n.json").read().strip().splitlines()[-1]); r = j["roofline"]; f = r["frame_kernels_ms"]
    print("%-10s N=%d  step %9.3f ms   render kernel(s) %9.3f ms   film gather %7.3f ms   rest of the step (zero, pack, resolve, launches) %7.3f ms   rays/frame %d"
          % ("$wl", $n, j["ms_per_step"], f["render"], f["film_gather"], j["ms_per_step"] - f["render"] - f["film_gather"], j["config"]["rays_per_frame"]))
except Exception as e:
    print("$wl N=$n FAILED", e); print(open("$OUT/${wl}_n$n.err").read()[-800:])
PY
  done
done
} 2>&1 | tee $OUT/emulate.txt
