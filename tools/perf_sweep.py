"""Build kernel variants (-D knobs) on the GPU box and bench each; prints one JSON line per variant."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HIP = os.path.join(ROOT, "pbrt-v1_amd", "csrc", "hip")
LIB = os.path.join(ROOT, "pbrt-v1_amd", "lib", "libpbrt_hip.so")
BASE = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value"]

def run(tag, defs=(), env=None, workload="c2", steps=2, extra=()):
    cmd = BASE + list(defs) + [os.path.join(HIP, "rt_kernels.hip"), os.path.join(HIP, "kd_build.cpp"), os.path.join(HIP, "grid_build.cpp"), "-o", LIB]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
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
    if which == "h":
        run("auto_c2"); run("c2_highocc", env={"PBRT_HIP_HIGH_OCC": "1"})
        run("auto_p100000", workload="p100000"); run("p100000_lowocc", env={"PBRT_HIP_HIGH_OCC": "0"}, workload="p100000")
        run("auto_c3_100000", workload="c3_100000"); run("c3_100000_lowocc", env={"PBRT_HIP_HIGH_OCC": "0"}, workload="c3_100000")
        return
    if which == "g":
        for wl in ("c2", "p100000"):
            run("mailbox_" + wl, workload=wl)
            run("nomailbox_" + wl, ["-DRT_MAILBOX=0"], workload=wl)
            run("w4_lds16_" + wl, ["-DRT_MIN_WAVES=4", "-DRT_STACK_LDS=16"], workload=wl)
            run("w3_lds16_" + wl, ["-DRT_STACK_LDS=16"], workload=wl)
            run("w5_lds12_" + wl, ["-DRT_MIN_WAVES=5", "-DRT_STACK_LDS=12"], workload=wl)
        return
    if which == "f":
        run("auto_c2")
        run("auto_c3_100k", workload="c3_100000")
        for mode in (0, 2):
            for ex in (16, 32, 40, 48, 56):
                run("c3_100k_mode%d_exit%d" % (mode, ex), env={"PBRT_HIP_TRAV_MODE": str(mode), "PBRT_HIP_EXIT_THRESH": str(ex)}, workload="c3_100000")
        run("c2_mode1_exit8", env={"PBRT_HIP_TRAV_MODE": "1", "PBRT_HIP_EXIT_THRESH": "8"})
        run("c2_mode1_exit16", env={"PBRT_HIP_TRAV_MODE": "1", "PBRT_HIP_EXIT_THRESH": "16"})
        return
    if which == "e":
        for k in (8, 16, 32, 48):
            run("batched_k%d" % k, ["-DRT_LOCKSTEP=2", "-DRT_BATCH_K=%d" % k])
            run("batched_k%d_c3_100k" % k, ["-DRT_LOCKSTEP=2", "-DRT_BATCH_K=%d" % k], workload="c3_100000")
        run("batched_k16_c3_100k_exit24", ["-DRT_LOCKSTEP=2", "-DRT_BATCH_K=16", "-DRT_EXIT_THRESH=24"], workload="c3_100000")
    if which == "d":
        run("lockstep")
        run("stepwise", ["-DRT_LOCKSTEP=0"])
        run("lockstep_c3_100k", workload="c3_100000")
        run("stepwise_c3_100k", ["-DRT_LOCKSTEP=0"], workload="c3_100000")
        run("lockstep_c3_100k_exit24", ["-DRT_EXIT_THRESH=24"], workload="c3_100000")
        run("lockstep_lds16", ["-DRT_STACK_LDS=16"])
    if which == "c":
        run("base")
        run("nofilm", env={"PBRT_HIP_DEBUG_NOFILM": "1"})
        run("c3_100k_base", workload="c3_100000")
        run("c3_100k_exit24", ["-DRT_EXIT_THRESH=24"], workload="c3_100000")
        run("c1", workload="c1", steps=5)
    if which == "b":
        run("base")
        run("nofilm", env={"PBRT_HIP_DEBUG_NOFILM": "1"})
        run("waves4_lds12", ["-DRT_MIN_WAVES=4", "-DRT_STACK_LDS=12"])
        run("waves4_lds16", ["-DRT_MIN_WAVES=4", "-DRT_STACK_LDS=16"])
        run("waves5_lds12", ["-DRT_MIN_WAVES=5", "-DRT_STACK_LDS=12"])
        run("waves4_lds12_exit16", ["-DRT_MIN_WAVES=4", "-DRT_STACK_LDS=12", "-DRT_EXIT_THRESH=16"])
        run("waves4_lds12_exit32", ["-DRT_MIN_WAVES=4", "-DRT_STACK_LDS=12", "-DRT_EXIT_THRESH=32"])
        run("c3_100k_base", workload="c3_100000")
        run("c3_100k_w4", ["-DRT_MIN_WAVES=4", "-DRT_STACK_LDS=16"], workload="c3_100000")
        run("c3_100k_w4_exit24", ["-DRT_MIN_WAVES=4", "-DRT_STACK_LDS=16", "-DRT_EXIT_THRESH=24"], workload="c3_100000")
    if which == "a":
        run("base")
        run("nofilm", env={"PBRT_HIP_DEBUG_NOFILM": "1"})
        run("waves2", ["-DRT_MIN_WAVES=2"])
        run("waves3_lds16", ["-DRT_MIN_WAVES=3", "-DRT_STACK_LDS=16"])
        run("waves4_lds12", ["-DRT_MIN_WAVES=4", "-DRT_STACK_LDS=12"])
        run("exit16", ["-DRT_EXIT_THRESH=16"])
        run("exit32", ["-DRT_EXIT_THRESH=32"])
        run("c3_100k_base", workload="c3_100000")
        run("c3_100k_exit24", ["-DRT_EXIT_THRESH=24"], workload="c3_100000")
    run("base_restore")

if __name__ == "__main__":
    main()
