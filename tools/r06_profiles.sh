#!/bin/bash
# GPU box, round 6: rocprofv3 passes for the five workloads (published locally afterwards with tools/publish_profile.py r06 <workload> --publish), C4 as stated with the
# scene-create log, emulated rank shares.
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_profiles; mkdir -p $O
for wl in c3 p1000000 c2 c4 c5; do ROUND=r06 tools/profile.sh $wl > $O/profile_$wl.log 2>&1; tail -2 $O/profile_$wl.log | cut -c1-200; done
PBRT_HIP_CREATE_LOG=1 python bench.py --workload c4full --no-cpu-baseline --no-extra --steps 2 --warmup 1 > $O/c4_full.json 2> $O/c4_full.err; grep -E "^CREATE|^KDBUILD" $O/c4_full.err | tee $O/scene_create_10m.txt; tail -c 400 $O/c4_full.json
tools/emulate_world.sh r06_emu "c3 p1000000"
