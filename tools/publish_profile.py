"""Summarise gpurun_out/prof_<round>_<tag>/ (tools/profile.sh) into summary.json there: per kernel {calls, total / average ms} from
--kernel-trace --stats and per-kernel PMC sums PER FRAME (the bench command renders 1 counted + 1 warm-up + 3 timed + 3 host-hand-over frames = 8).
`--publish` also writes profiles/<round>_<tag>{_kernel_stats.csv,_summary.json} and, for the workload's dominant timed kernel,
profiles/<round>_<tag>_render_kernel.json -- what bench.py reads `roofline.traffic` / `valu_issue` from through the latest_<workload>
symlink.  Every published file carries `code_id` = pbrt_v1_amd.code_id() of the library that was profiled (sha256 of its device code):
bench.py prints the counters only when the library it has loaded has the same one.
usage: python tools/publish_profile.py <round> <tag> [--publish]"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
rnd, tag = sys.argv[1], sys.argv[2]
publish = "--publish" in sys.argv[3:]
src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (rnd, tag))
wl = tag.split("_pipe")[0]
FRAMES = 8          # bench.py --steps 3 --warmup 1: 1 counted (counting twin) + 1 warm-up + 3 timed + 3 host-hand-over frames
code_id = entry.load_package().code_id()
out = {"workload": wl, "round": rnd, "code_id": code_id,
       "command": "rocprofv3 --kernel-trace --stats | --pmc <group> (one group per run) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload " + wl,
       "frames_per_run": FRAMES, "kernels": {}, "pmc_per_frame": {}}
KERNELS = ("render_kernel", "pipe_trace", "pipe_shade", "pipe_vertex", "pipe_march", "film_slot", "film_march", "film_gather", "derive_")
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("void ", "")
        if any(k in n for k in KERNELS):
            out["kernels"][n] = {"calls": int(r["Calls"]), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "avg_ms": round(float(r["AverageNs"]) / 1e6, 4),
                                 "ms_per_frame_all_calls": round(float(r["TotalDurationNs"]) / 1e6 / FRAMES, 3)}
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "?").split("(")[0].replace("void ", "")
        if any(k in n for k in KERNELS):
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, d in acc.items():
    timed = "<false" in n
    film = "film_" in n                                                                       # the film gather runs once in every frame
    out["pmc_per_frame"][n] = {c: v / (FRAMES if film else FRAMES - 1 if timed else 1) for c, v in d.items()}     # the counting twin renders one frame, the timed kernel four
    p = out["pmc_per_frame"][n]
    dv = {}
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        dv["fabric_bytes_lower_bound"] = (p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
        dv["fabric_bytes_fetch_doubled"] = (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
    if p.get("TCC_HIT_sum", 0) + p.get("TCC_MISS_sum", 0) > 0:
        dv["L2_hit_rate"] = p["TCC_HIT_sum"] / (p["TCC_HIT_sum"] + p["TCC_MISS_sum"])
        dv["L2_misses_per_frame"] = p["TCC_MISS_sum"]
    if p.get("SQ_ACTIVE_INST_VALU") and p.get("GRBM_GUI_ACTIVE"):
        # profiles/valu_issue_calibration.json: SQ_ACTIVE_INST_VALU counts 1 per full-rate instruction (2 per quarter-rate one), a wave64 instruction
        # occupies its SIMD-32 for 2 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        dv["VALUBusy_percent"] = 100.0 * p["SQ_ACTIVE_INST_VALU"] * 2 / 1024 / (p["GRBM_GUI_ACTIVE"] / 8)
    if p.get("SQ_THREAD_CYCLES_VALU") and p.get("SQ_ACTIVE_INST_VALU"):
        dv["lanes_active_percent"] = 100.0 * p["SQ_THREAD_CYCLES_VALU"] / (p["SQ_ACTIVE_INST_VALU"] * 64)
    if p.get("SQ_LDS_BANK_CONFLICT") and p.get("SQ_ACTIVE_INST_LDS"):
        dv["LDS_bank_conflict_cycles_per_active_LDS_cycle"] = p["SQ_LDS_BANK_CONFLICT"] / p["SQ_ACTIVE_INST_LDS"]
    out["pmc_per_frame"][n]["derived"] = dv
json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:5000])

if publish:
    prof = os.path.join(ROOT, "profiles")
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(prof, "%s_%s_kernel_stats.csv" % (rnd, tag)))
    json.dump(out, open(os.path.join(prof, "%s_%s_summary.json" % (rnd, tag)), "w"), indent=1)
    # dominant timed kernel = the timed (COUNT=false) kernel with the most time per frame
    timed = {n: k for n, k in out["kernels"].items() if "<false" in n and any(x in n for x in ("render_kernel", "pipe_trace", "pipe_march"))}
    name = max(timed, key=lambda n: timed[n]["total_ms"])
    p = dict(out["pmc_per_frame"][name]); d = p.pop("derived")
    frames_timed = FRAMES - 1
    rk = {"round": rnd, "workload": wl, "kernel": name, "code_id": code_id, "command": out["command"],
          "ms_per_frame_kernel_trace": round(timed[name]["total_ms"] / frames_timed, 3), "calls": timed[name]["calls"],
          "FETCH_SIZE_KB": p["FETCH_SIZE"], "WRITE_SIZE_KB": p["WRITE_SIZE"],
          "hbm_bytes_per_launch_uncorrected": d["fabric_bytes_lower_bound"], "hbm_bytes_per_launch_fetch_doubled": d["fabric_bytes_fetch_doubled"],
          "note_traffic": "per FRAME (all launches of the kernel in one frame); MI355X_MICROARCH.md HBM section: bytes = (FETCH_SIZE + WRITE_SIZE)*1024; "
                          "FETCH_SIZE tallies 64 B per fabric read request (profiles/r01_fetch_size_calibration.txt), hence the doubled upper bound",
          "pmc": p,
          "derived": {"VALUBusy_percent": d.get("VALUBusy_percent"), "VALUUtilization_percent_active_lanes": d.get("lanes_active_percent"),
                      "L2_hit_rate": d.get("L2_hit_rate"), "L2_misses_per_frame": d.get("L2_misses_per_frame")}}
    json.dump(rk, open(os.path.join(prof, "%s_%s_render_kernel.json" % (rnd, tag)), "w"), indent=1)
    if tag == wl:
        link = os.path.join(prof, "latest_%s_render_kernel.json" % wl)
        if os.path.lexists(link): os.remove(link)
        os.symlink("%s_%s_render_kernel.json" % (rnd, wl), link)
