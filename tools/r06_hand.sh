#!/bin/bash
# GPU box, round 6: lane hand-over in the by-vertex megakernel (RT_MEGA_HANDOVER) -- parity, timing, rank-0 shares at N = 8, lane occupancy counters
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
V=${1:-hand}
STEPS=6 WARMUP=2 tools/ab_scan.sh r06_${V}_scan "$V default" "p1000000 c4 c2" "test"
tools/emulate_world.sh r06_${V}_emu "p1000000" $V
tools/emulate_world.sh r06_${V}_emu_default "p1000000"
O=$GRAFT_REPO_ROOT/gpurun_out/r06_${V}_scan
cd /tmp && export TMPDIR=/tmp
for lib in $V default; do
  [ $lib != default ] && export PBRT_HIP_LIB_PATH=$GRAFT_REPO_ROOT/pbrt-v1_amd/lib/libpbrt_hip_$lib.so || unset PBRT_HIP_LIB_PATH
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_$lib -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload p1000000 > $O/pmc_$lib.log 2>&1
  python - <<PY | tee -a $O/scan.txt
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("$O/pmc_$lib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel<false" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
if acc:
    print("$lib p1000000: lanes active (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64) = %.1f %%   SQ_INSTS_VALU %.4g" % (100 * acc["SQ_THREAD_CYCLES_VALU"] / max(acc["SQ_ACTIVE_INST_VALU"], 1) / 64, acc["SQ_INSTS_VALU"]))
PY
  rm -rf $O/pmc_$lib
done
