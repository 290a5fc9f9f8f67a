"""GPU box: rebuild single units with -D knobs and time workloads.  usage: python tools/r03_scan.py <group>"""
import json, os, subprocess, sys
os.environ.setdefault("PBRT_HIP_TUNE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import r02_trace_scan as T
T.UNITS = ("rt_kernels", "rt_trace", "rt_mega_w", "rt_mega_d", "rt_mega_p", "rt_pipe_w", "rt_pipe_d", "rt_pipe_p", "rt_pipe_v", "kd_build", "grid_build")

def bench(tag, env=None, workload="p1000000", steps=3):
    e = dict(env or {}); e.setdefault("PBRT_HIP_PIPELINE", "")
    T.bench(tag, env=e, workload=workload, steps=steps)

g = sys.argv[1]
if g == "occ":
    for w in (5, 6, 4):
        for u in ("rt_mega_p", "rt_mega_d"):
            T.rebuild(u, ["-DRT_HIGH_OCC_WAVES=%d" % w])
        for wl in ("c3", "p1000000", "c4"):
            bench("mega_waves%d_%s" % (w, wl), workload=wl)
elif g == "exit":
    for et in (8, 16, 24, 32, 40, 48, 56):
        for wl in ("c3", "p1000000"):
            bench("exit%d_%s" % (et, wl), env={"PBRT_HIP_EXIT_THRESH": str(et)}, workload=wl)
elif g == "trace8":
    # the trace kernel at 8 waves per SIMD (64 VGPRs) with a 6-entry LDS ring (18 KB per workgroup: 8 workgroups per CU)
    for defs in (["-DRT_TRACE_WAVES=8", "-DRT_TRACE_STACK=6"], ["-DRT_TRACE_WAVES=7", "-DRT_TRACE_STACK=6"], []):
        T.rebuild("rt_trace", defs)
        for wl in ("c5", "p1000000"):
            bench("trace:%s:%s" % (" ".join(defs), wl), env={"PBRT_HIP_PIPELINE": "1"}, workload=wl)

elif g == "steps":
    for ds, lm in ((2, 24), (3, 24), (6, 24), (4, 16), (4, 32), (3, 16), (4, 24)):
        for u in ("rt_mega_p", "rt_mega_d"):
            T.rebuild(u, ["-DRT_TRACE_DSTEPS=%d" % ds, "-DRT_TRACE_LEAF_MIN=%d" % lm])
        for wl in ("c3", "p1000000"):
            bench("dsteps%d_leafmin%d_%s" % (ds, lm, wl), workload=wl)

elif g == "steps2":
    for ds, lm in ((1, 24), (2, 16), (2, 32), (2, 24)):
        for u in ("rt_mega_p", "rt_mega_d", "rt_trace"):
            T.rebuild(u, ["-DRT_TRACE_DSTEPS=%d" % ds, "-DRT_TRACE_LEAF_MIN=%d" % lm])
        for wl in ("c3", "p1000000"):
            bench("dsteps%d_leafmin%d_%s" % (ds, lm, wl), workload=wl)
        bench("dsteps%d_leafmin%d_c5" % (ds, lm), workload="c5")

elif g == "shadeocc":
    for w in (3, 4, 1):
        T.rebuild("rt_pipe_d", ["-DRT_SHADE_VOL_WAVES=%d" % w])
        bench("shade_vol_waves%d_c5" % w, workload="c5")

elif g == "prof":
    for u in ("rt_mega_p", "rt_mega_d", "rt_kernels"):
        T.rebuild(u, ["-DRT_PROFILE"])
    for wl in ("c3", "p1000000", "c2"):
        e = dict(os.environ); e["PBRT_HIP_PIPELINE"] = "0"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "1", "--warmup", "0", "--workload", wl], env=e, capture_output=True, text=True, timeout=600)
        print(wl, "\n".join([l for l in r.stderr.splitlines() if l.startswith("RT_PROFILE")][-1:]), flush=True)

elif g == "stages":
    # per-stage cycle budget of the megakernel (-DRT_PROFILE -DRT_PROFILE_STAGES: s_memtime around every stage of the state machine)
    for u in ("rt_mega_p", "rt_mega_d", "rt_kernels"):
        T.rebuild(u, ["-DRT_PROFILE", "-DRT_PROFILE_STAGES"])
    for wl in ("c2", "c3", "p1000000"):
        e = dict(os.environ); e["PBRT_HIP_PIPELINE"] = "0"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "1", "--warmup", "0", "--workload", wl], env=e, capture_output=True, text=True, timeout=600)
        print("==", wl, flush=True)
        print("\n".join([l for l in r.stderr.splitlines() if l.startswith("RT_PROFILE")][-20:]), flush=True)
        if r.returncode: print(r.stderr[-600:], flush=True)
elif g == "region":
    for m in (2, 4, 6, 0):
        for u in ("rt_mega_p", "rt_mega_d", "rt_trace"):
            T.rebuild(u, ["-DRT_STEP_REGION=%d" % m])
        for wl in ("c3", "p1000000"):
            bench("region%d_%s" % (m, wl), workload=wl)
        bench("region%d_c5" % m, workload="c5")

elif g == "c5final":
    # C5 on the 32 M-slot pool: steps per round of the trace kernel, waves of the volume shade kernels, leaf batch
    for defs in (["-DRT_PIPE_TRACE_DSTEPS=2"], ["-DRT_TRACE_LEAF_MIN=16"], ["-DRT_TRACE_LEAF_MIN=32"], []):
        T.rebuild("rt_trace", defs)
        bench("c5final_trace_%s" % "_".join(defs), workload="c5", steps=2)
    for w in (2, 4, 3):
        T.rebuild("rt_pipe_d", ["-DRT_SHADE_VOL_WAVES=%d" % w])
        bench("c5final_shadewaves%d" % w, workload="c5", steps=2)
elif g == "roundform":
    # the round's remaining compile-time choices on the final build: pop inside the descent loop, leaving the descent steps once N lanes hold a primitive
    for defs in (["-DRT_TRACE_POP_IN_LOOP=1"], ["-DRT_TRACE_LEAF_GO=16"], ["-DRT_TRACE_LEAF_GO=32"], ["-DRT_TRACE_DSTEPS=3"], []):
        T.rebuild("rt_mega_p", defs)
        for wl in ("c2", "p1000000"):
            bench("roundform_%s_%s" % ("_".join(defs), wl), workload=wl, steps=2)
elif g == "c2steps":
    # C2 on the 4-wave flavour with batched rounds: steps per round and leaf batch (tuned on the 1 M-triangle frames so far)
    for ds, lm in ((1, 24), (2, 8), (1, 8), (3, 24), (2, 24)):
        T.rebuild("rt_mega_p", ["-DRT_TRACE_DSTEPS=%d" % ds, "-DRT_TRACE_LEAF_MIN=%d" % lm])
        bench("c2steps_d%d_l%d" % (ds, lm), workload="c2", steps=2)
elif g == "c2occ":
    # C2 on the register-capped flavour (batched rounds, exit threshold 8) at 4 / 5 / 6 / 3 waves per SIMD
    for w in (5, 6, 3, 4):
        T.rebuild("rt_mega_p", ["-DRT_HIGH_OCC_WAVES=%d" % w])
        for et in (8, 12):
            bench("c2_occwaves%d_et%d" % (w, et), env={"PBRT_HIP_HIGH_OCC": "1", "PBRT_HIP_TRAV_MODE": "2", "PBRT_HIP_EXIT_THRESH": str(et)}, workload="c2")
elif g == "slot2":
    for pf, un in ((8, 8), (4, 8), (10, 8)):
        T.rebuild("rt_kernels", ["-DRT_SLOT_PF=%d" % pf, "-DRT_SLOT_UNROLL=%d" % un])
        for wl in ("c2", "c3"):
            tag = "slot_pf%d_un%d_%s" % (pf, un, wl)
            bench(tag, workload=wl, steps=2)
            try:
                j = json.load(open(os.path.join(ROOT, "gpurun_out", "scan_" + tag + ".json")))
                print("   ", tag, "film_gather ms", j["roofline"]["frame_kernels_ms"]["film_gather"], flush=True)
            except Exception as ex:
                print("   ", tag, ex, flush=True)
elif g == "slot":
    # film_slot_kernel ablations on C2 / C3 (the gather's time is in roofline.frame_kernels_ms.film_gather)
    for m in (1, 2, 4, 6, 7, 0):
        T.rebuild("rt_kernels", ["-DRT_SLOT_PROBE=%d" % m])
        for wl in ("c2", "c3"):
            bench("slotprobe%d_%s" % (m, wl), workload=wl, steps=2)
            try:
                j = json.load(open(os.path.join(ROOT, "gpurun_out", "scan_slotprobe%d_%s.json" % (m, wl))))
                print("   probe", m, wl, "film_gather ms", j["roofline"]["frame_kernels_ms"]["film_gather"], flush=True)
            except Exception as ex:
                print("   probe", m, wl, ex, flush=True)
