#!/bin/bash
# GPU box: which unit binds a traversal kernel -- TA / TCP / SQ counters of one bench.py workload, one rocprofv3 --pmc pass per group.
#   tools/pmc_probe.sh TAG WORKLOAD [LIB_VARIANT]          -> gpurun_out/<TAG>/counters.txt (per frame: sums over the 3 timed + 1 warm-up frames / 4)
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
TAG=$1; WL=$2; VAR=${3:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
[ -n "$VAR" ] && export PBRT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_$VAR.so
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload $WL"
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE TD_TD_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o t -- $CMD > $OUT/pmc$i.log 2>&1 || tail -3 $OUT/pmc$i.log
done
python - <<PY | tee $OUT/counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "<false" in n and ("render_kernel" in n or "pipe_" in n): acc[n][r["Counter_Name"]] += float(r["Counter_Value"]) / 4
print("workload $WL, library ${VAR:-product}; per frame")
for n, d in acc.items():
    print(n)
    for k in sorted(d): print("   %-45s %.4g" % (k, d[k]))
PY
rm -rf $OUT/pmc*/
