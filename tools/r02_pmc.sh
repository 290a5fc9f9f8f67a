#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# GPU box: PMC passes over the pipeline's kernels for one workload.  usage: tools/r02_pmc.sh <tag> <workload> [env...]
TAG=$1; WL=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL"
rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE TCC_BUSY_avr TCC_TAG_STALL_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o t -- $CMD > $OUT/pmc$i.log 2>&1
done
python - <<PY
import csv, glob, json, os
from collections import defaultdict
out = "$OUT"
res = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    res["kernel_stats"] = [r for r in csv.DictReader(open(f))][:10]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r.get("Kernel_Name", "?")[:48]][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0)))
res["pmc"] = {k: {c: {"n": len(v), "sum": sum(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, d in res["pmc"].items():
    if "trace" in k or "render_kernel" in k or "shade" in k:
        print(k); [print("   %-34s n=%5d sum=%.4g mean=%.4g" % (c, v["n"], v["sum"], v["mean"])) for c, v in sorted(d.items())]
for r in res.get("kernel_stats", []): print(r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"))
PY
grep -il "error\|invalid" $OUT/pmc*.log | head
