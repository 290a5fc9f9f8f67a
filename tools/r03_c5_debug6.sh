export PBRT_HIP_TUNE=1   # the library reads its PBRT_HIP_* knobs only then
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
p = e.load_package(); p.build(defines=("-fno-slp-vectorize",))
PY
timeout 300 python tools/r03_c5_debug4.py mega 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload c2 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 noslp', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"
timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload c3 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 noslp', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"
timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload p1000000 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p1m noslp', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"
