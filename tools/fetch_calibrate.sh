#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# usage (GPU box): bash tools/fetch_calibrate.sh  -> gpurun_out/fetch_calibrate.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/fetch_calibrate.hip -o /tmp/fetch_calibrate || exit 1
timeout 300 /tmp/fetch_calibrate > $OUT/known.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o t -- /tmp/fetch_calibrate > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o t -- /tmp/fetch_calibrate > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/rdreq -o t -- /tmp/fetch_calibrate > $OUT/rdreq.log 2>&1
python - <<PY
import csv, glob
print(open("$OUT/known.txt").read())
for sub in ("fetch", "write", "rdreq"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            print(sub, r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"])
PY
