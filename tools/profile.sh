#!/bin/bash
# GPU box: rocprofv3 evidence for one bench.py workload.  usage: [ROUND=r05] [PROF_TAG=_x] tools/profile.sh <workload> [--publish]
# Pass 1: --kernel-trace --stats.  Passes 2..: PMC counters, each group in its own run (no tracing domains next to --pmc).
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
set -u
WL=$1; shift
ROUND=${ROUND:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${ROUND}_$WL${PROF_TAG:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload $WL"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o t -- $CMD > $OUT/pmc$i.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/publish_profile.py $ROUND $WL${PROF_TAG:-} "$@"
