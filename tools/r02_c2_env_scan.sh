#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# GPU box: the C2 headline frame under the megakernel's run-time scheduling knobs
run() { env "$@" python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 2>/dev/null | TAG="$*" python -c '
import json, os, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({"env": os.environ["TAG"], "ms": j["ms_per_step"], "kernel_ms": j["roofline"]["kernel_ms"]}))'; }
run X=1
for t in 0 8 16 24 48; do run PBRT_HIP_EXIT_THRESH=$t; done
run PBRT_HIP_PHASE_SYNC=0
run PBRT_HIP_PHASE_SYNC=1
run PBRT_HIP_TRAV_MODE=2
run PBRT_HIP_TRAV_MODE=4
run PBRT_HIP_HIGH_OCC=1
