#!/bin/bash
export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
# GPU box: rank 0's share of an N-rank job on one GPU (no collective), N = 1 2 4 8: what the per-rank kernels cost at each N
for n in 1 2 4 8; do
  PBRT_BENCH_EMULATE_WORLD=$n python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 1 ${1:+--workload $1} 2>/dev/null | N=$n python -c '
import json, os, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({"emulated_world": int(os.environ["N"]), "workload": j["config"]["workload"][:40], "rank0_ms_per_step": j["ms_per_step"], "kernel_ms": j["roofline"]["kernel_ms"], "frame_kernels_ms": j["roofline"]["frame_kernels_ms"]}))'
done
