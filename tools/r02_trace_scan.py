"""GPU box: rebuild only rt_trace.hip with -D knobs, relink, and time the 1 M-triangle path frame through the queue pipeline.
usage: python tools/r02_trace_scan.py <group>"""
import json, os, subprocess, sys
os.environ.setdefault("PBRT_HIP_TUNE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HIP = os.path.join(ROOT, "pbrt-v1_amd", "csrc", "hip")
LIBD = os.path.join(ROOT, "pbrt-v1_amd", "lib")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-Wno-unused-value"]
UNITS = ("rt_kernels", "rt_trace", "rt_mega_w", "rt_mega_d", "rt_mega_p", "rt_pipe_w", "rt_pipe_d", "rt_pipe_p", "kd_build", "grid_build")


def rebuild(unit, defs):
    src = os.path.join(HIP, unit + (".cpp" if unit.endswith("build") else ".hip"))
    subprocess.check_call(["hipcc"] + FLAGS + list(defs) + ["-c", src, "-o", os.path.join(LIBD, "obj", unit + ".o")], stderr=subprocess.DEVNULL)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(LIBD, "obj", u + ".o") for u in UNITS] +
                          ["-o", os.path.join(LIBD, "libpbrt_hip.so")])


def bench(tag, env=None, workload="p1000000", steps=3):
    e = dict(os.environ); e.update(env or {}); e.setdefault("PBRT_HIP_PIPELINE", "1")
    if e["PBRT_HIP_PIPELINE"] == "": del e["PBRT_HIP_PIPELINE"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", str(steps), "--warmup", "1", "--workload", workload],
                       env=e, capture_output=True, text=True, timeout=400)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1]); ro = j["roofline"]
        print(json.dumps(dict(tag=tag, workload=workload, Mrays=j["value"], ms=j["ms_per_step"], trace_ms=ro["kernel_ms"], frac=ro["frac"],
                              nodes_per_ray=ro["nodes_per_ray"], tris_per_ray=ro["tri_tests_per_ray"], render_ms=ro["frame_kernels_ms"]["render"], iters=ro["pipeline_iterations"], slots=ro["pipeline_slots"])), flush=True)
        open(os.path.join(ROOT, "gpurun_out", "scan_" + tag.replace(":", "_").replace(" ", "_") + ".json"), "w").write(json.dumps(j))
    except Exception as ex:
        print(json.dumps(dict(tag=tag, error=str(ex), stderr=r.stderr[-400:])), flush=True)


def main():
    g = sys.argv[1] if len(sys.argv) > 1 else "occ"
    if g == "occ":
        for w in (8, 6, 5, 4, 3):
            rebuild("rt_trace", ["-DRT_TRACE_WAVES=%d" % w])
            bench("waves%d" % w)
        rebuild("rt_trace", ["-DRT_TRACE_WAVES=8", "-DRT_TRACE_REFILL=32"]); bench("waves8_refill32")
        rebuild("rt_trace", ["-DRT_TRACE_WAVES=8", "-DRT_TRACE_REFILL=8"]); bench("waves8_refill8")
        rebuild("rt_trace", ["-DRT_TRACE_WAVES=8", "-DRT_BATCH_K=8"]); bench("waves8_batch8")
        rebuild("rt_trace", ["-DRT_TRACE_WAVES=8", "-DRT_BATCH_K=32"]); bench("waves8_batch32")
        rebuild("rt_trace", [])
        for sl in (1 << 20, 1 << 23):
            bench("slots%d" % sl, env={"PBRT_HIP_PIPE_SLOTS": str(sl)})
    elif g == "loop":
        for L in (0, 1, 2):
            rebuild("rt_trace", ["-DRT_TRACE_LOOP=%d" % L, "-DRT_TRACE_REFILL=32"]); bench("loop%d_refill32" % L)
        for bpc in (8, 6, 4, 3, 2):
            bench("loop2_blocks%d" % bpc, env={"PBRT_HIP_TRACE_BLOCKS_PER_CU": str(bpc)})
        for ds, lm in ((2, 12), (8, 12), (4, 4), (4, 24), (16, 24), (64, 1)):
            rebuild("rt_trace", ["-DRT_TRACE_REFILL=32", "-DRT_TRACE_DSTEPS=%d" % ds, "-DRT_TRACE_LEAF_MIN=%d" % lm]); bench("loop2_dsteps%d_leafmin%d" % (ds, lm))
        rebuild("rt_trace", ["-DRT_TRACE_REFILL=48"]); bench("loop2_refill48")
        rebuild("rt_trace", ["-DRT_TRACE_REFILL=64"]); bench("loop2_refill64")
        rebuild("rt_trace", [])
    elif g == "slots":
        for sl in (1 << 22, 1 << 23, 1 << 24, 17 << 20):
            bench("slots%d" % sl, env={"PBRT_HIP_PIPE_SLOTS": str(sl)})
            bench("slots%d_blocks5" % sl, env={"PBRT_HIP_PIPE_SLOTS": str(sl), "PBRT_HIP_TRACE_BLOCKS_PER_CU": "5"})
        for sl in (1 << 23, 17 << 20):
            bench("c3_slots%d" % sl, env={"PBRT_HIP_PIPE_SLOTS": str(sl)}, workload="c3")
    elif g == "mega":
        for wl in ("p1000000", "c3", "p100000", "c4", "c5"):
            bench("mega_flat_" + wl, env={"PBRT_HIP_PIPELINE": "0"}, workload=wl)
            bench("mega_batched_" + wl, env={"PBRT_HIP_PIPELINE": "0", "PBRT_HIP_TRAV_MODE": "4"}, workload=wl)
            bench("pipe8M_" + wl, env={"PBRT_HIP_PIPELINE": "1", "PBRT_HIP_PIPE_SLOTS": str(1 << 23)}, workload=wl)
    elif g == "c2":
        for defs in ([], ["-DRT_POOLED_BRANCHY_DESCENT"]):
            for u in ("rt_mega_p", "rt_mega_d", "rt_mega_w"):
                rebuild(u, defs)
            for wl in ("c2", "c2d", "c2w"):
                bench("c2scan:" + " ".join(defs), env={"PBRT_HIP_PIPELINE": "0"}, workload=wl)
        for ds, lm in ((2, 12), (8, 12), (4, 4), (4, 24), (8, 32)):
            rebuild("rt_mega_p", ["-DRT_TRACE_DSTEPS=%d" % ds, "-DRT_TRACE_LEAF_MIN=%d" % lm])
            bench("mega_dsteps%d_leafmin%d" % (ds, lm), env={"PBRT_HIP_PIPELINE": "0"}, workload="p1000000")
        rebuild("rt_mega_p", [])
        for et in (16, 24, 40, 48):
            bench("mega_exit%d" % et, env={"PBRT_HIP_PIPELINE": "0", "PBRT_HIP_EXIT_THRESH": str(et)}, workload="p1000000")
    elif g == "occ2":
        for wl in ("p1000000", "c3", "c4"):
            bench("mega_highocc1_" + wl, env={"PBRT_HIP_PIPELINE": "0", "PBRT_HIP_HIGH_OCC": "1"}, workload=wl)
            bench("mega_highocc0_" + wl, env={"PBRT_HIP_PIPELINE": "0", "PBRT_HIP_HIGH_OCC": "0"}, workload=wl)
    elif g == "prof":
        rebuild("rt_mega_p", ["-DRT_PROFILE"]); rebuild("rt_kernels", ["-DRT_PROFILE"])
        for wl in ("p1000000", "c2"):
            e = dict(os.environ); e["PBRT_HIP_PIPELINE"] = "0"
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "1", "--warmup", "0", "--workload", wl], env=e, capture_output=True, text=True, timeout=400)
            print(wl, "\n".join([l for l in r.stderr.splitlines() if l.startswith("RT_PROFILE")][-2:]), flush=True)
        rebuild("rt_mega_p", []); rebuild("rt_kernels", [])
    elif g == "stack":
        for ns in (16, 8):
            for u in ("rt_mega_p", "rt_mega_d"):
                rebuild(u, ["-DRT_STACK_LDS=%d" % ns])
            for wl in ("p1000000", "c3", "c2"):
                bench("mega_stack%d_%s" % (ns, wl), env={"PBRT_HIP_PIPELINE": "0"}, workload=wl)
        for u in ("rt_mega_p", "rt_mega_d"):
            rebuild(u, [])
    elif g == "util3":
        # slot categories of the descent steps (probe build: nodes = step slots, leaf_refs = of which the lane is idle, spills = of which the
        # lane holds an untested primitive, tris = test slots), then timings of the same knobs without the probe
        sets = [[], ["-DRT_TRACE_POP_IN_LOOP=1"], ["-DRT_TRACE_REFILL=16"], ["-DRT_TRACE_REFILL=8"], ["-DRT_TRACE_DSTEPS=8"], ["-DRT_TRACE_DSTEPS=2"],
                ["-DRT_TRACE_LEAF_MIN=12"], ["-DRT_TRACE_LEAF_MIN=40"], ["-DRT_TRACE_DSTEPS=16", "-DRT_TRACE_LEAF_GO=24"], ["-DRT_TRACE_DSTEPS=16", "-DRT_TRACE_LEAF_GO=32", "-DRT_TRACE_LEAF_MIN=16"]]
        for defs in sets:
            rebuild("rt_trace", defs + ["-DRT_PROBE_UTIL"])
            e = dict(os.environ); e.update({"PBRT_HIP_PIPELINE": "1", "PBRT_BENCH_COUNTERS": "1"})
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "1", "--warmup", "0", "--workload", "p1000000"], env=e, capture_output=True, text=True, timeout=400)
            c = [json.loads(l[9:]) for l in r.stderr.splitlines() if l.startswith("COUNTERS ")][-1]
            rays = c["closest_rays"] + c["any_rays"]
            print(json.dumps(dict(tag="probe:" + " ".join(defs), step_slots=round(c["nodes_visited"] / rays, 1), idle=round(c["leaf_refs"] / rays, 1), leaf_pending=round(c["stack_overflows"] / rays, 1), test_slots=round(c["tri_tests"] / rays, 1))), flush=True)
            rebuild("rt_trace", defs)
            bench("time:" + " ".join(defs), steps=2)
        rebuild("rt_trace", [])
    elif g == "chunk":
        for defs in ([], ["-DRT_TRACE_REFILL=8"], ["-DRT_TRACE_REFILL=24"], ["-DRT_TRACE_REFILL=32"], ["-DRT_TRACE_CHUNK_MAX=128"], ["-DRT_TRACE_CHUNK_MAX=512"],
                     ["-DRT_TRACE_REFILL=8", "-DRT_TRACE_CHUNK_MAX=512"], ["-DRT_TRACE_REFILL=4", "-DRT_TRACE_CHUNK_MAX=512"]):
            rebuild("rt_trace", defs)
            bench("chunk:" + " ".join(defs), steps=2)
        rebuild("rt_trace", [])
        bench("chunk:default c5", workload="c5", steps=2); bench("chunk:default c3", workload="c3", steps=2)
    elif g == "fold":
        for defs in (["-DRT_TRACE_FOLD=0"], []):
            rebuild("rt_trace", defs); rebuild("rt_mega_p", defs); rebuild("rt_mega_d", defs)
            for wl in ("p1000000", "c5"):
                bench("pipe fold:" + " ".join(defs), workload=wl, steps=2)
            for wl in ("p1000000", "c3", "c4", "p100000"):
                bench("mega fold:" + " ".join(defs), env={"PBRT_HIP_PIPELINE": "0"}, workload=wl, steps=2)
    elif g == "occ3":
        for w in (3, 5, 6):
            for u in ("rt_mega_p", "rt_mega_d"):
                rebuild(u, ["-DRT_HIGH_OCC_WAVES=%d" % w])
            for wl in ("p1000000", "c3", "c4"):
                bench("mega highocc_waves%d" % w, env={"PBRT_HIP_PIPELINE": "0"}, workload=wl, steps=2)
        for u in ("rt_mega_p", "rt_mega_d"):
            rebuild(u, [])
    elif g == "treelet":
        # order of the sibling-pair records: depth-first vs breadth-first treelets of T records, with / without line alignment
        for T, al in ((1, 0), (4, 0), (8, 0), (8, 1), (16, 0), (16, 1), (32, 1)):
            e = {"PBRT_HIP_TREELET_PAIRS": str(T), "PBRT_HIP_TREELET_ALIGN": str(al)}
            bench("treelet%d_align%d mega" % (T, al), env=dict(e, PBRT_HIP_PIPELINE="0"), steps=2)
            bench("treelet%d_align%d pipe" % (T, al), env=dict(e, PBRT_HIP_PIPELINE="1"), steps=2)
    elif g == "treelet2":
        for T, al in ((1, 0), (8, 1)):
            e = {"PBRT_HIP_TREELET_PAIRS": str(T), "PBRT_HIP_TREELET_ALIGN": str(al)}
            for wl in ("c5", "c3", "c4", "p100000"):
                bench("treelet%d_align%d default-arch" % (T, al), env=dict(e, PBRT_HIP_PIPELINE=""), workload=wl, steps=3)
    elif g == "gather":
        # where the film gather's time goes on the C2 frame: staging only / accumulation only / both
        for defs in (["-DRT_GATHER_NOACC"], ["-DRT_GATHER_NOSTAGE"], []):
            rebuild("rt_kernels", defs)
            e = dict(os.environ)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "3", "--warmup", "1"], env=e, capture_output=True, text=True, timeout=400)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            print(json.dumps(dict(tag="gather:" + " ".join(defs), ms=j["ms_per_step"], kernels=j["roofline"]["frame_kernels_ms"])), flush=True)
    elif g == "megachunk":
        for ch in (128, 256):
            rebuild("rt_mega_p", ["-DRT_MEGA_CHUNK=%d" % ch])
            for wl in ("c2", "p1000000"):
                bench("megachunk%d" % ch, env={"PBRT_HIP_PIPELINE": "0"}, workload=wl, steps=2)
        rebuild("rt_mega_p", [])
        for wl in ("c2", "p1000000", "c3", "c4"):
            bench("megachunk64", env={"PBRT_HIP_PIPELINE": "0"}, workload=wl, steps=2)
    elif g == "util":
        # lane slots per ray (every lane of a wave that executes a step / a leaf test counts) next to the useful visits
        for defs in (["-DRT_PROBE_UTIL"], []):
            rebuild("rt_trace", defs)
            for wl in ("p1000000", "c5", "c3"):
                bench("util:" + " ".join(defs), workload=wl, steps=1)
    elif g == "defs":
        defs = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else []
        rebuild("rt_trace", defs)
        for wl in sys.argv[3:] or ["p1000000"]:
            bench("defs:" + " ".join(defs), workload=wl)


if __name__ == "__main__":
    main()
