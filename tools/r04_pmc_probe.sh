#!/bin/bash
# which unit binds the 1 M-triangle megakernel: TA / TCP / SQ counters of one p1000000 run each
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA|TCP|TD|SQ|TCC|GRBM)_[A-Z0-9_]+" | sort -u > $OUT/avail.txt
wc -l $OUT/avail.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload ${WL:-p1000000}"
i=0
for set in "TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o t -- $CMD > $OUT/pmc$i.log 2>&1 || tail -3 $OUT/pmc$i.log
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "<false" in n and ("render_kernel" in n or "pipe_" in n): acc[n][r["Counter_Name"]] += float(r["Counter_Value"]) / 4
for n, d in acc.items():
    print(n)
    for k in sorted(d): print("   %-45s %.4g" % (k, d[k]))
PY
