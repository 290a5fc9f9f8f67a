#!/bin/bash
# round 4: tile-ordered work list in the single-shard megakernel (PBRT_HIP_MEGA_TILE = tile edge in pixels, 0 = scanline order)
export PBRT_HIP_TUNE=1
ulimit -c 0
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_probe9; mkdir -p $OUT
PBRT_HIP_MEGA_TILE=32 timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "flavour or 1m_direct or 1m_path or c3_full or golden" 2>&1 | tail -5 | tee $OUT/tests.txt
grep -q " passed" $OUT/tests.txt && ! grep -q "failed" $OUT/tests.txt || { echo "parity tests failed: no timings"; exit 1; }
run() {  # tag, workload, env...
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 1 --workload $wl > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "Mrays/s", j["ms_per_step"], "ms/frame", "kernel_ms", r["kernel_ms"], "frac", r["frac"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-1500:])
PY
}
{
for wl in c3 p1000000 c4 c2; do
  for t in 0 8 16 32 64; do
    run ${wl}_tile$t $wl PBRT_HIP_MEGA_TILE=$t
  done
done
run c3_tile16_nobands c3 PBRT_HIP_MEGA_TILE=16 PBRT_HIP_XCD_BANDS=0
run c3_tile32_nobands c3 PBRT_HIP_MEGA_TILE=32 PBRT_HIP_XCD_BANDS=0
run p1000000_tile16_bands p1000000 PBRT_HIP_MEGA_TILE=16 PBRT_HIP_XCD_BANDS=1
run p1000000_tile32_bands p1000000 PBRT_HIP_MEGA_TILE=32 PBRT_HIP_XCD_BANDS=1
} 2>&1 | tee $OUT/scan.txt
