"""Build kernel variants (-D knobs) on the GPU box and bench each; prints one JSON line per variant."""
import json, os, subprocess, sys
os.environ.setdefault("PBRT_HIP_TUNE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HIP = os.path.join(ROOT, "pbrt-v1_amd", "csrc", "hip")
LIB = os.path.join(ROOT, "pbrt-v1_amd", "lib", "libpbrt_hip.so")
BASE = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value"]

_built = None


def run(tag, defs=(), env=None, workload="c2", steps=2, extra=()):
    global _built
    if _built == tuple(defs):
        return _bench(tag, defs, env, workload, steps, extra)
    _built = tuple(defs)
    cmd = BASE + list(defs) + [os.path.join(HIP, "rt_kernels.hip"), os.path.join(HIP, "kd_build.cpp"), os.path.join(HIP, "grid_build.cpp"), "-o", LIB]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return _bench(tag, defs, env, workload, steps, extra)


def _bench(tag, defs, env, workload, steps, extra):
    e = dict(os.environ); e.update(env or {})
    try:
      r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", str(steps), "--warmup", "1",
                        "--workload", workload] + list(extra), env=e, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
      print(json.dumps(dict(tag=tag, error="timeout"))); return
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(json.dumps(dict(tag=tag, workload=workload, Mrays=j["value"], ms=j["ms_per_step"], kernel_ms=j["roofline"]["kernel_ms"],
                              frac=j["roofline"]["frac"], rays_per_frame=j["config"]["rays_per_frame"])), flush=True)
    except Exception as ex:
        print(json.dumps(dict(tag=tag, error=str(ex), stderr=r.stderr[-300:])), flush=True)
    if "-DRT_PROFILE" in defs:
        print("\n".join(l for l in r.stderr.splitlines() if l.startswith("RT_PROFILE")[-3:]) if False else "\n".join([l for l in r.stderr.splitlines() if l.startswith("RT_PROFILE")][-14:]), flush=True)

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "a"
    if which == "u":
        for wl in ("p1000000", "c3", "p100000", "c2"):
            run("pairfetch_" + wl, ["-DRT_PAIR_FETCH"], workload=wl, steps=3)
        return
    if which == "t":
        for k in (8, 32):
            run("batch_k%d_p1m" % k, ["-DRT_BATCH_K=%d" % k], workload="p1000000")
            run("batch_k%d_c3" % k, ["-DRT_BATCH_K=%d" % k], workload="c3")
        return
    if which == "s":
        for w in (4, 5):
            run("highocc%d_p1m" % w, ["-DRT_HIGH_OCC_WAVES=%d" % w], workload="p1000000")
            run("highocc%d_p100k" % w, ["-DRT_HIGH_OCC_WAVES=%d" % w], workload="p100000")
        run("lowocc_p1m", env={"PBRT_HIP_HIGH_OCC": "0"}, workload="p1000000")
        run("lowocc_p100k", env={"PBRT_HIP_HIGH_OCC": "0"}, workload="p100000")
        return
    if which == "r":
        for w in (5, 6, 7, 8):
            run("highocc%d_p1m" % w, ["-DRT_HIGH_OCC_WAVES=%d" % w], workload="p1000000")
            run("highocc%d_p100k" % w, ["-DRT_HIGH_OCC_WAVES=%d" % w], workload="p100000")
        run("highocc8_lds10_p1m", ["-DRT_HIGH_OCC_WAVES=8", "-DRT_STACK_LDS=10"], workload="p1000000")
        return
    if which == "q":
        run("gather_full"); run("gather_nostage", ["-DRT_GATHER_NOSTAGE"]); run("gather_noacc", ["-DRT_GATHER_NOACC"])
        run("gather_none", ["-DRT_GATHER_NOACC", "-DRT_GATHER_NOSTAGE"]); run("nofilm", env={"PBRT_HIP_DEBUG_NOFILM": "1"})
        return
    if which == "p":
        for wl in ("c2",):
            run("profile_" + wl, ["-DRT_PROFILE"], workload=wl)
            run("profile_stages_" + wl, ["-DRT_PROFILE", "-DRT_PROFILE_STAGES"], workload=wl)
        return
    raise SystemExit("usage: perf_sweep.py p|q|r|s|t  (p: RT_PROFILE cycle split, q: film-gather ablation, r/s: occupancy flavours on the soups)")


if __name__ == "__main__":
    main()
