#!/bin/bash
# round 4, first GPU call: (1) the VALU issue ceiling + what the SQ counters count per instruction, (2) fused traversal round A/B,
# (3) the parity tests that exercise the traversal on the 1 M-triangle tree
export PBRT_HIP_TUNE=1
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe1; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/valu_issue_bench.hip -o /tmp/valu_issue_bench 2>/dev/null || exit 1
  timeout 300 /tmp/valu_issue_bench 20000 > $OUT/valu_issue.txt 2>&1
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $set | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -o t -- /tmp/valu_issue_bench 20000 4 > $OUT/pmc_$tag.log 2>&1
  done
  python - <<PY
import csv, glob, collections
print(open("$OUT/valu_issue.txt").read())
agg = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r.get("Dispatch_Id"), r["Counter_Name"])
        agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
for k, v in agg.items(): print("PMC", k[0], "dispatch", k[1], k[2], v)
PY
) > $OUT/valu_summary.txt 2>&1
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame trace_ms", r["kernel_ms"], "render_ms", r["frame_kernels_ms"]["render"], "shade", r["frame_kernels_ms"]["shade_launches"], "gather", r["frame_kernels_ms"]["film_gather"], "iters", r.get("pipeline_iterations"), "frac", r["frac"], "frac_frame", r["frac_frame_kernels"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
{
for wl in p1000000 c3 c4; do
  for f in 0 1 2; do run ${wl}_fused$f $wl PBRT_HIP_FUSED=$f; done
done
run c2_fused0 c2 PBRT_HIP_FUSED=0
run c2_fused1 c2 PBRT_HIP_FUSED=1
run c5_fusedtrace c5
run c5_nofuse c5 PBRT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_nofuse.so
run p1m_pipe_fused p1000000 PBRT_HIP_PIPELINE=1
run p1m_pipe_nofuse p1000000 PBRT_HIP_PIPELINE=1 PBRT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_nofuse.so
} 2>&1 | tee $OUT/scan.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "trace or 1m or pair or flavour" 2>&1 | tail -15 | tee $OUT/tests.txt
