#!/bin/bash
# GPU box: the round's final evidence in one call -- GPU tests, the full bench line, rocprofv3 passes for the five workloads (published locally afterwards with
# tools/publish_profile.py r06 <workload> --publish), emulated rank shares, C4 as stated with the scene-create log.
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_final; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
( time python bench.py > $O/bench_full.json 2> $O/bench_full.err ) 2> $O/bench_time.txt; tail -c 600 $O/bench_full.json; cat $O/bench_time.txt
for wl in c3 c2 p1000000 c4 c5; do ROUND=r06 tools/profile.sh $wl > $O/profile_$wl.log 2>&1; tail -2 $O/profile_$wl.log | cut -c1-200; done
[ -z "$NO_EMU" ] && tools/emulate_world.sh r06_emu "c3 p1000000"
PBRT_HIP_CREATE_LOG=1 python bench.py --workload c4full --no-cpu-baseline --no-extra --steps 2 --warmup 1 > $O/c4_full.json 2> $O/c4_full.err; grep -E "^CREATE|^KDBUILD" $O/c4_full.err | tee $O/scene_create_10m.txt; tail -c 400 $O/c4_full.json
[ -z "$NO_EMU" ] && STEPS=2 tools/emulate_world.sh r06_emu_c4full "c4full"
