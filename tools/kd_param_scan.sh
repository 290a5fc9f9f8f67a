#!/bin/bash
# GPU box: KdTreeAccel's build parameters (accelerators/kdtree.cpp:489-498) scanned on the device -- the same frames on different trees.
#   tools/kd_param_scan.sh TAG "WORKLOADS" -> gpurun_out/<TAG>/scan.txt     (each line: intersectcost traversalcost emptybonus maxprims | nodes, build s, ms/frame, Mrays/s)
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
TAG=$1; WORKLOADS=${2:-"p1000000 c3"}
SETS=${SETS:-"80:1:0.5:1 20:1:0.5:1 8:1:0.5:1 4:1:0.5:1 2:1:0.5:1 1:1:0.5:1 8:1:0.5:2 4:1:0.5:2 2:1:0.5:2 8:1:0.5:4 4:1:0.5:4 2:1:0.5:4 4:1:0.2:2 4:1:0:2 4:1:0.8:2 2:1:0.5:8"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
{
for wl in $WORKLOADS; do
  for set in $SETS; do
    IFS=: read ic tc eb mp <<< "$set"
    export PBRT_BENCH_TUNED_ACCEL="\"integer intersectcost\" [$ic] \"integer traversalcost\" [$tc] \"float emptybonus\" [$eb] \"integer maxprims\" [$mp]"
    tag=${wl}_$(echo $set | tr ':.' '__')
    timeout 600 python bench.py --no-cpu-baseline --no-extra --steps ${STEPS:-4} --warmup 1 --workload ${wl}_tuned > $OUT/$tag.json 2> $OUT/$tag.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]; c = j["config"]
    print("%-12s isect %3s trav %s eb %-4s maxprims %s | %10d nodes  build %6.2f s | %8.3f ms/frame  kernel %8.3f ms  %8.1f Mrays/s  frac %.3f  nodes/ray %6.1f tris/ray %5.1f B/ray %6.0f"
          % ("$wl", "$ic", "$tc", "$eb", "$mp", c["kd_nodes"], c["kd_build_s"], j["ms_per_step"], r["kernel_ms"], j["value"], r["frac"], r["nodes_per_ray"], r["tri_tests_per_ray"], r["bytes_per_ray"]))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-600:])
PY
  done
done
} 2>&1 | tee $OUT/scan.txt
