#!/usr/bin/env python3
"""bench.py -- Mrays/s of the pbrt-v1 Scene::Render hot path on N MI355X GPUs (one process per GPU).

A *step* is one full frame of the workload.  The headline is BASELINE.json configs[2], the largest configuration stated for ONE
GPU ("1M-triangle synthetic trianglemesh (random soup), DirectLighting, 1920x1080 @ 16 spp, 1 MI355X"); configs[1] (Cornell, path
depth 5, 1024x1024 @ 64 spp: cache-resident, VALU-bound) and the other configs at single-GPU size are sub-records of the same line.
Every camera sample goes through camera-ray generation, kd-tree traversal + triangle intersection, the radiance estimate and the
filtered film splat, then (N > 1) the RCCL merge of the film accumulators, and ImageFilm::WriteImage's normalisation: the timed step ends with the
RESOLVED film in device memory; copying it to page-locked host memory (the caller's PCIe transfer) is timed in a second loop and reported as
`host_handover`, never as `value`.  Rays = every Scene::Intersect + every
Scene::IntersectP call (camera, bounce, MIS closest-hit and shadow rays), the metric's definition.
Scene data is resident in HBM before the timed region; the frame is fixed, so N > 1 is strong scaling
(image tiles dealt round-robin to ranks).

Prints ONE JSON line on rank 0 (see the contract in the task description) including
  roofline     : algorithmic bytes (8 B/node visit + 4 B/leaf ref + 48 B/triangle test + 48 B/ray,
                 SURVEY.md section 8d) of one frame / the HIP-event duration of the dominant kernel
                 (rt::render_kernel, or the rt::pipe_trace_kernel launches of a frame summed), against
                 8 TB/s HBM; `gather_ceiling` = the same work priced in L2-missing line requests at the
                 56 G/s this chip sustains; `traffic` from the committed rocprofv3
                 PMC summary of the same command (profiles/)
  cpu_baseline : the *reference itself* (oracle/_ref/pbrt_ref, built from /root/reference with its own
                 flags, single thread, its own MT19937 stream) timed on this box's host cores on a
                 centre crop window of the same frame
  per_rank     : kernel / render ms and rays of every rank (load balance)
  workloads    : (N = 1, default run) full sub-records -- value, ms_per_step, roofline, cpu_baseline --
                 for the other BASELINE configs at the size one GPU holds: C2, the 1 M-triangle path frame
                 (the north star's case), C4 (1 M triangles, material mix, path depth 8), C5 (SURVEY 8d inputs: 64 spp, stepsize 20, g 0).
                 `roofline.frac` is priced on the dominant kernel alone; `frac_frame_kernels` on every kernel of the frame.
                 (N > 1) the configurations BASELINE.json states for several GPUs, each with `per_rank`: C4 as stated (10 M triangles,
                 2048 x 2048 @ 256 spp) at 8 ranks, C5 at 4 and 8 ranks (--multi-workloads).
N > 1 merge per frame: rt_film_pack_parts, ONE reduce-scatter (chunk r = the 5 planes of rank r's film rows), rt_film_resolve_device_rgba
on those rows, ONE all-gather of RGBA rows.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


HEADLINE = "c3"

# KdTreeAccel's build parameters (accelerators/kdtree.cpp:489-498: intersectcost 80, traversalcost 1, emptybonus 0.5, maxprims 1, maxdepth -1) are tuned for the
# reference's CPU; SURVEY.md section 7: "keep defaults for parity runs, allow tuned params for perf runs, report both".  The `<workload>_tuned` sub-records render the
# SAME frame on the tree these parameters give (closest hits do not depend on the tree: tests/test_gpu_configs.py::test_tuned_tree_gives_the_same_film) and their
# reference-CPU leg runs the same text.  Chosen by tools/kd_param_scan.py on the MI355X (profiles/r06_kd_param_scan.txt).
TUNED_ACCEL = os.environ.get("PBRT_BENCH_TUNED_ACCEL") or \
    '"integer intersectcost" [%d] "integer traversalcost" [%d] "float emptybonus" [%s] "integer maxprims" [%d]' % (2, 1, "0", 4)


def workload(name: str):
    from pbrt_v1_amd import scenes
    if name.endswith("_tuned"):
        text, label, crop = workload(name[:-len("_tuned")])
        text = accel_with_params(text, TUNED_ACCEL)
        return text, label + " -- kd-tree built with " + TUNED_ACCEL.replace('"', ""), crop
    if name == "c2":
        text = scenes.cornell_scene(xres=1024, yres=1024, integrator="path", maxdepth=5, xsamples=8, ysamples=8,
                                    jitter=True, pixel_filter="mitchell", accelerator="kdtree")
        label = "Cornell box (12 tris + 2-tri area light), PathIntegrator maxdepth=5, 1024x1024 @ 64 spp (stratified 8x8 jittered), mitchell 2x2 filter, kd-tree"
        crop = (0.34, 0.66, 0.34, 0.66)                       # ~10 % of the frame: ~14 s of reference CPU time
    elif name in ("c2w", "c2d"):        # same frame as c2 with the cheaper integrators (where does the time go?)
        integ = "whitted" if name == "c2w" else "directlighting"
        text = scenes.cornell_scene(xres=1024, yres=1024, integrator=integ, xsamples=8, ysamples=8, jitter=True, pixel_filter="mitchell")
        label = "Cornell box, %s, 1024x1024 @ 64 spp, mitchell, kd-tree" % integ
        crop = (0.4375, 0.5625, 0.4375, 0.5625)
    elif name == "tsmall":              # tests/test_multirank_gpu.py
        text = scenes.cornell_scene(xres=160, yres=120, integrator="path", maxdepth=5, xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell", soup_tris=3000)
        label = "test frame: Cornell + 3000-triangle soup, path, 160x120 @ 4 spp"
        crop = (0.4, 0.6, 0.4, 0.6)
    elif name == "t8":                  # tests/test_multirank_gpu.py: C4's kind of frame (material mix, path depth 8) at a size 8 ranks on one GPU render in seconds;
        text = scenes.cornell_scene(xres=640, yres=426, integrator="path", maxdepth=8, xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell",   # 426 rows: not a multiple of 8
                                    soup_tris=3000, soup_materials=True)
        label = "test frame: Cornell + 3000-triangle soup, matte/glass/mirror mix, path depth 8, 640x426 @ 4 spp"
        crop = (0.4, 0.6, 0.4, 0.6)
    elif name == "t5":                  # tests/test_multirank_gpu.py: C5's kind of frame (homogeneous medium, single scattering: the queue pipeline with the march kernel)
        text = scenes.cornell_scene(xres=320, yres=214, integrator="directlighting", xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell", soup_tris=3000,
                                    volume_integrator='"single" "float stepsize" [20]', world_kwargs=dict(volume='"float g" [0]'))
        label = "test frame: Cornell + 3000-triangle soup in a homogeneous medium, single scattering + DirectLighting, 320x214 @ 4 spp"
        crop = (0.4, 0.6, 0.4, 0.6)
    elif name == "c1":
        text = scenes.cornell_scene(xres=512, yres=512, integrator="whitted", xsamples=1, ysamples=1, jitter=False,
                                    pixel_filter="box")
        label = "Cornell box, WhittedIntegrator, 512x512 @ 1 spp, box filter, kd-tree"
        crop = None
    elif name.startswith("c3"):
        ntri = 1_000_000 if name == "c3" else int(name.split("_")[1])
        text = scenes.cornell_scene(xres=1920, yres=1080, integrator="directlighting", xsamples=4, ysamples=4,
                                    jitter=True, pixel_filter="mitchell", soup_tris=ntri)
        label = "Cornell + %d-triangle LCG soup, DirectLighting(all), 1920x1080 @ 16 spp, mitchell, kd-tree" % ntri
        crop = (0.2, 0.8, 0.2, 0.8)
    elif name == "c4full":                  # BASELINE config 4 as stated: 10 M triangles, 2048 x 2048 @ 256 spp (1.07 G camera samples per frame, a 34 GB sample
        ntri = 10_000_000                   # buffer: fits one MI355X; its reference-CPU leg would take days, so this one is a committed record: profiles/r04_c4_full.json)
        text = scenes.cornell_scene(xres=2048, yres=2048, integrator="path", maxdepth=8, xsamples=16, ysamples=16, jitter=True,
                                    pixel_filter="mitchell", soup_tris=ntri, soup_materials=True)
        label = "Cornell + %d-triangle LCG soup, matte/glass/mirror mix, PathIntegrator maxdepth=8, 2048x2048 @ 256 spp, mitchell, kd-tree" % ntri
        crop = (0.495, 0.505, 0.495, 0.505)
    elif name.startswith("c4"):             # c4_1000000: BASELINE config 4 at single-GPU size (material mix, path depth 8)
        ntri = 1_000_000 if name == "c4" else int(name.split("_")[1])
        text = scenes.cornell_scene(xres=1024, yres=1024, integrator="path", maxdepth=8, xsamples=4, ysamples=4, jitter=True,
                                    pixel_filter="mitchell", soup_tris=ntri, soup_materials=True)
        label = "Cornell + %d-triangle LCG soup, matte/glass/mirror mix, PathIntegrator maxdepth=8, 1024x1024 @ 16 spp, mitchell, kd-tree" % ntri
        crop = (0.39, 0.61, 0.39, 0.61)
    elif name.startswith("c5"):             # c5_1000000: BASELINE config 5 at single-GPU size (homogeneous medium, single scattering)
        ntri = 1_000_000 if name == "c5" else int(name.split("_")[1])
        # sigma_a = sigma_s = 0.002, g = 0, Le = 0, stepsize 20, 64 spp
        text = scenes.cornell_scene(xres=1024, yres=1024, integrator="directlighting", xsamples=8, ysamples=8, jitter=True, pixel_filter="mitchell",
                                    soup_tris=ntri, volume_integrator='"single" "float stepsize" [20]', world_kwargs=dict(volume='"float g" [0]'))
        label = "Cornell + %d-triangle LCG soup in a homogeneous medium (sigma_a = sigma_s = .002, g = 0), single-scattering volume integrator (stepsize 20) + DirectLighting, 1024x1024 @ 64 spp" % ntri
        crop = (0.45, 0.55, 0.45, 0.55)
    elif name.startswith("p"):              # p1000000: Cornell + N-triangle soup, path tracing (the north-star's 1M-triangle case)
        ntri = int(name[1:])
        text = scenes.cornell_scene(xres=1024, yres=1024, integrator="path", maxdepth=5, xsamples=4, ysamples=4,
                                    jitter=True, pixel_filter="mitchell", soup_tris=ntri)
        label = "Cornell + %d-triangle LCG soup, PathIntegrator maxdepth=5, 1024x1024 @ 16 spp, mitchell, kd-tree" % ntri
        crop = (0.39, 0.61, 0.39, 0.61)
    else:
        raise SystemExit("unknown workload " + name)
    return text, label, crop


def accel_with_params(text: str, params: str) -> str:
    """the scene text with `params` appended to its Accelerator line (options block only: a world can be a gigabyte of triangles)"""
    import re
    i = text.find("WorldBegin")
    head = text[:i]
    new = re.sub(r'(Accelerator "\w+")[^\n]*', lambda m: m.group(1) + " " + params, head)
    assert new != head, "no Accelerator line"
    return new + text[i:]


def oracle_side_text(name: str, crop, keyed: bool):
    """The workload's scene as the compiled reference is given it: centre crop window of the same frame, its accelerator wrapped in the ray-counting helper
    plugin, and (keyed) its sampler wrapped in the keyed-RNG helper (oracle/ref; pbrt_v1_amd.scenes.for_product unwraps both again for the product)."""
    import re
    text, _, _ = workload(name)
    i = text.find("WorldBegin")
    head = text[:i]
    if crop is not None:
        head = head.replace('"string filename"', '"float cropwindow" [%s %s %s %s] "string filename"' % crop)
    head = re.sub(r'Accelerator "(\w+)"', r'Accelerator "countaccel" "string inner" ["\1"]', head)
    if keyed:
        head = re.sub(r'Sampler "(\w+)"', r'Sampler "keyed" "string inner" ["\1"] "integer seed" [0]', head)
    return head + text[i:]


def ref_runner():
    """oracle/ref_runner.py (test infrastructure: runs the compiled reference in oracle/_ref)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pbrt_ref_runner", os.path.join(ROOT, "oracle", "ref_runner.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def cpu_baseline(pkg, name: str, crop, budget_s: float = 25.0):
    """Time the compiled reference (single thread) on a crop window of the same frame."""
    text = oracle_side_text(name, crop, keyed=False)
    try:
        t0 = time.time()
        _, _, st = ref_runner().run_reference(text, keyed=False, timeout=600)
        rays = st["closest_rays"] + st["any_rays"]
        return {"value": round(rays / st["render_s"] / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": "oracle/_ref/pbrt_ref (reference built with its own flags -O2 -msse2 -mfpmath=sse, 1 thread, MT19937) on "
                          "cropwindow %s of the same frame: %d rays in %.2f s render (+%.3f s kd build); host has %d cores"
                          % (crop, rays, st["render_s"], st["accel_build_s"], os.cpu_count()),
                "wall_s": round(time.time() - t0, 2)}
    except FileNotFoundError:
        return {"value": None, "unit": "Mrays/s", "cores": 1, "kind": "port", "sample": "oracle/_ref missing on this box"}


def parity_leg(pkg, name: str, crop, device_index: int):
    """The third leg of BASELINE.json's metric -- per-pixel L2 against the reference's film -- at the workload's own size: the crop window the cpu_baseline leg
    times is rendered on the device as a frame of its own BY THE TIMED KERNELS (COUNT = false: what `value` is measured on) and compared with the compiled reference
    under the keyed RNG (oracle/_ref/pbrt_ref_keyed: same sample seeds) on the same text.  The reference's film and Scene::Intersect(P) counts come from
    oracle/ (the checker); everything compared with them comes from libpbrt_hip.so.  film/image.cpp:157-212 is what both sides end with."""
    import numpy as np
    text = oracle_side_text(name, crop, keyed=True)
    try:
        t0 = time.time()
        ref_rgb, ref_alpha, st = ref_runner().run_reference(text, keyed=True, timeout=900)
        ref_s = time.time() - t0
    except FileNotFoundError:
        return {"value": None, "reason": "oracle/_ref missing on this box"}
    ps = pkg.ParsedScene(text=text)                          # (for_product unwraps the two helper plugins)
    ds = pkg.DeviceScene(ps, device=device_index)
    try:
        ds.set_counting(False); ds.render(); rgb, alpha = ds.film()
        ds.set_counting(True); ds.reset_counters(); ds.clear_film(); ds.render(); rgb_c, alpha_c = ds.film(); cnt = ds.counters()
    finally:
        ds.close()
    if rgb.shape != ref_rgb.shape:
        return {"value": None, "reason": "film shapes differ: device %s, reference %s" % (rgb.shape, ref_rgb.shape)}
    l2 = np.sqrt(((rgb.astype(np.float64) - ref_rgb) ** 2).sum(-1))
    rays_dev = cnt["closest_rays"] + cnt["any_rays"]; rays_ref = st["closest_rays"] + st["any_rays"]
    return {"crop": list(crop) if crop is not None else None, "pixels": int(l2.size), "film": [int(rgb.shape[1]), int(rgb.shape[0])],
            "max_abs": float(np.abs(rgb - ref_rgb).max()), "max_l2": float(l2.max()), "mean_l2": float(l2.mean()),
            "frac_within_1e-4": float((l2 < 1e-4).mean()), "alpha_max_abs": float(np.abs(alpha - ref_alpha).max()),
            "ray_count_equal": bool(rays_dev == rays_ref), "rays_device": int(rays_dev), "rays_reference": int(rays_ref),
            # (path frames: a bounce direction that differs in its last bit -- device libm -- now and then changes what a later ray meets; DirectLighting frames: two
            # triangles hit at EXACTLY equal t -- the reference's per-primitive mailbox, accelerators/kdtree.cpp:376-386, and this library's re-test keep different
            # ones when the first was already tested in an earlier leaf: 2 of 12 M camera rays on C3's crop, tests/probe_equal_t_ties.py, profiles/r06_tie_probe.txt)
            "ray_count_rel_diff": float(abs(rays_dev - rays_ref) / max(rays_ref, 1)),
            "timed_kernel_film_equals_counting_twin": bool(np.array_equal(rgb, rgb_c) and np.array_equal(alpha, alpha_c)),
            "tolerance": "per-pixel L2 over rgb < 1e-4 (north star); DirectLighting / Whitted frames are expected bit-exact, path frames >= 99.5 % of pixels within 1e-4 (device libm in the cosine-sampled bounce)",
            "reference": "oracle/_ref/pbrt_ref_keyed (compiled reference, keyed RNG, seed 0) on the same text, %.1f s" % ref_s}


def attach_profile(out, name, alg_bytes, loaded, profiles_dir=None):
    """roofline.traffic / valu_issue from the committed rocprofv3 PMC summary of this workload (tools/profile.sh), but only when that profile was
    collected on the device code this process has loaded (pbrt_v1_amd.code_id()); a stale profile is reported as stale."""
    profiles_dir = profiles_dir or os.path.join(ROOT, "profiles")
    prof = os.path.join(profiles_dir, "latest_%s_render_kernel.json" % name)
    if os.path.exists(prof):
        pj = json.load(open(prof))
        if pj.get("code_id") != loaded:
            # the counters were collected on other kernels than the ones this process has loaded (a kernel changed and tools/profile.sh was not
            # re-run): say so instead of printing stale counters next to fresh times
            out["roofline"]["traffic_source"] = ("profiles/%s is stale: collected on device code %s, this process loaded %s -- re-run tools/profile.sh %s --publish"
                                                 % (os.path.basename(os.path.realpath(prof)), pj.get("code_id", "(none recorded)"), loaded, name))
        else:
            # profiles/r01_fetch_size_calibration.txt: on this library's gathers FETCH_SIZE tallies 64 B per fabric read request
            # (exact for 64-B requests, 1/2 for 128-B ones, the guide's streaming case); WRITE_SIZE is exact.  `traffic` is the
            # conservative figure (reads doubled); the lower bound is reported beside it.
            out["roofline"]["traffic"] = int(pj["hbm_bytes_per_launch_fetch_doubled"])
            out["roofline"]["traffic_lower_bound"] = int(pj["hbm_bytes_per_launch_uncorrected"])
            out["roofline"]["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per frame; device code %s = the loaded library)" % (os.path.basename(os.path.realpath(prof)), loaded)
            d = pj.get("derived", {})
            # VALU issue of the same kernel: wave-level VALU instructions per frame (SQ_INSTS_VALU of the same profile) against the issue rate
            # MEASURED on this chip (tools/valu_issue_bench.hip, profiles/valu_issue_calibration.json): 1.02e12 wave64 instructions/s for
            # plain VALU streams at 6-8 waves per SIMD (2.4 cycles each; the guide's 2 cycles = 1.2288e12 is reached by no stream).  Compares,
            # lane-mask selects and scalar mask arithmetic -- half of a select-style traversal step -- run at 0.585e12 when not interleaved,
            # so a kernel of that mix saturates below 1.0 (C2, cache-resident, reaches ~0.72).
            iv = pj.get("pmc", {}).get("SQ_INSTS_VALU")
            if iv:
                cal = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_calibration.json")))
                peak = float(cal["bench_py_uses"]["valu_issue.peak_wave_insts_per_s"])
                out["roofline"]["valu_issue"] = {"wave_insts_per_launch": int(iv), "peak_wave_insts_per_s": peak, "peak_source": "profiles/valu_issue_calibration.json (measured)",
                                                 "ms_at_peak_issue": round(iv / peak * 1e3, 2),
                                                 "frac": round(iv / peak * 1e3 / max(out["roofline"]["kernel_ms"], 1e-9), 4),
                                                 "lanes_active": round(d.get("VALUUtilization_percent_active_lanes", 0) / 100.0, 3)}
            busy = "VALUBusy %.0f%%, %.0f%% of lanes active" % (d.get("VALUBusy_percent", 0), d.get("VALUUtilization_percent_active_lanes", 0))
            if out["roofline"]["traffic"] < 0.1 * alg_bytes:
                out["roofline"]["note"] = ("this workload's tree and triangles are L1/L2 resident: measured HBM traffic is ~%.0f%% of the algorithmic "
                                           "bytes (mostly the 32-byte sample records written once), so the HBM roofline is not the binding limit "
                                           "here; VALU issue under divergence is (%s)" % (100.0 * out["roofline"]["traffic"] / alg_bytes, busy))
            else:
                out["roofline"]["note"] = ("fabric traffic %.2f-%.2fx the algorithmic bytes (one 64-byte request per gather that misses L2; L2 hit rate %.0f%%); %s"
                                           % (out["roofline"]["traffic_lower_bound"] / alg_bytes, out["roofline"]["traffic"] / alg_bytes, 100 * d.get("L2_hit_rate", 0), busy))


def run_workload(name, args, pkg, torch, dist, world, rank, device_index, steps, warmup, with_cpu, dump_film=None):
    """Render `steps` timed frames of one workload on this rank's shard; rank 0 returns the record (others None)."""
    import numpy as np
    text, label, crop = workload(name)
    ps = pkg.ParsedScene(text=text)
    if not ps.valid or ps.errors:
        raise SystemExit("bench.py: the workload's scene description did not parse cleanly (%d errors): refusing to time a different scene" % ps.errors)
    emu = int(os.environ.get("PBRT_BENCH_EMULATE_WORLD", "0"))      # debugging aid: time rank 0's share of an N-rank job on one GPU
    tiles = (args.tile_2d, args.tile_h if args.tile_h > 0 else args.tile_2d) if args.tile_2d > 0 else args.tile_pixels
    tiles_used = ps.set_shard(rank, emu if (emu > 1 and world == 1) else world, tiles, fit=True)   # 2-D tiles sized to pad the sample extent least
    # One accelerator build per node, not per rank: local rank 0 builds (all host cores: 10 M triangles take 15 s) and publishes the
    # flattened tree under /dev/shm; the other ranks map it (rt_scene_create_prebuilt).  Reference: every cropwindow process builds its own.
    share = dist is not None and not os.environ.get("PBRT_BENCH_NO_SHARED_ACCEL")
    t_create = time.perf_counter()
    if not share:
        ds = pkg.DeviceScene(ps, device=device_index)
    else:
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        where = [None]
        if local_rank == 0:
            ds = pkg.DeviceScene(ps, device=device_index)
            for d in ("/dev/shm", "/tmp"):                                # the first place with room for it (10 M triangles: 6 GB); none: every rank builds its own
                path = os.path.join(d, "pbrt_hip_accel_%s_%s.bin" % (os.environ.get("MASTER_PORT", "0"), name))
                try:
                    pkg.publish_accel(ds, path)
                    where[0] = path
                    break
                except OSError as e:
                    print("bench.py: cannot publish the accelerator under %s (%s)" % (d, e), file=sys.stderr, flush=True)
        # every node's local rank 0 publishes on its own (it may have chosen another directory than global rank 0, or failed): a rank takes the path of ITS
        # node's local rank 0, never one that exists only on another host
        import socket
        mine = (socket.gethostname(), local_rank, where[0])
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        shared = next((p for h, lr, p in everyone if h == mine[0] and lr == 0), None)
        if local_rank != 0:
            ds = pkg.DeviceScene(ps, device=device_index, prebuilt=pkg.attach_accel(shared)) if shared else pkg.DeviceScene(ps, device=device_index)
        dist.barrier()
        if local_rank == 0 and shared:
            os.remove(shared)
    t_create = time.perf_counter() - t_create
    info = ds.accel_info()
    film = torch.zeros((5, ps.height, ps.width), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    ds.set_stream(stream.cuda_stream)
    ds.bind_film(film.data_ptr())
    # the host side of the boundary: the resolved film lands in page-locked buffers that are reused every frame
    host_rgb_t = torch.empty((ps.height, ps.width, 3), dtype=torch.float32, pin_memory=True) if rank == 0 else None
    host_alpha_t = torch.empty((ps.height, ps.width), dtype=torch.float32, pin_memory=True) if rank == 0 else None
    host_rgb = host_rgb_t.numpy() if rank == 0 else None
    host_alpha = host_alpha_t.numpy() if rank == 0 else None

    # N > 1: how the partial films meet (film/image.cpp:220-228 + tools/exrassemble.cpp:42-133 in the reference).
    #   allreduce       one all-reduce(sum) of the 5 planes, rank 0 resolves the whole film;
    #   reduce_scatter  (default) row-wise reduce-scatter per plane, every rank resolves ITS rows (rt_film_resolve_device), one all-gather of
    #                   the resolved rows (4 floats per pixel instead of 5, 1/N of the normalisation per rank).
    merge = args.merge if dist is not None else "none"
    emulate = emu if (emu > 1 and world == 1) else 0            # one GPU renders rank 0's share of an `emu`-rank job AND runs the merge's own kernels (no collective)
    if emulate and merge == "none":
        merge = "reduce_scatter"
    mworld = emulate or world
    H, W = ps.height, ps.width
    if merge == "reduce_scatter":
        # ONE reduce-scatter + ONE all-gather per frame (round 5; rounds 3-4: one reduce-scatter per plane, two all-gathers and a padding copy):
        # rt_film_pack_parts lays the rank's film out as `world` parts of `rows` film rows, part r = [5][rows][W] = everything rank r resolves;
        # rt_film_resolve_device_rgba turns the summed part into RGBA rows, which one all-gather hands to everybody.
        rows = (H + mworld - 1) // mworld
        send = torch.zeros((mworld, 5, rows, W), dtype=torch.float32, device="cuda")
        part = torch.zeros((5, rows, W), dtype=torch.float32, device="cuda")
        rgba_part = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda")
        rgba_all = torch.zeros((rows * mworld, W, 4), dtype=torch.float32, device="cuda")
        host_rgba_t = torch.empty((H, W, 4), dtype=torch.float32, pin_memory=True) if rank == 0 else None
        if rank == 0:
            host_rgb, host_alpha = host_rgba_t.numpy()[..., :3], host_rgba_t.numpy()[..., 3]
    host_backend = dist is not None and dist.get_backend() != "nccl"      # the 2-ranks-on-one-GPU test harness: collectives through the host

    # The timed step ends with the RESOLVED film in device memory (ImageFilm::WriteImage's arithmetic done, rt_film_resolve_device); handing it to host
    # memory is the caller's PCIe transfer, measured separately below (`host_handover`) and never part of `value`.
    if merge != "reduce_scatter":
        rgb_dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); alpha_dev = torch.zeros((H, W), dtype=torch.float32, device="cuda")

    def step():
        film.zero_()
        ds.render(sync=False)
        if merge == "allreduce":
            dist.all_reduce(film, op=dist.ReduceOp.SUM)
        elif merge == "reduce_scatter":
            ds.pack_parts(film.data_ptr(), mworld, rows, send.data_ptr())
            if emulate:
                part.copy_(send[0])                                  # what the reduce-scatter would leave here, minus the other ranks' samples
            elif host_backend:
                ho = torch.zeros((5, rows, W))
                dist.reduce_scatter_tensor(ho.view(-1), send.cpu().view(-1))
                part.copy_(ho)
            else:
                dist.reduce_scatter_tensor(part.view(-1), send.view(-1))      # (flat views: chunk r of `send` is part r)
            ds.resolve_device_rgba(part.data_ptr(), rows * W, rgba_part.data_ptr())
            if emulate:
                rgba_all[:rows].copy_(rgba_part)
            elif host_backend:
                hr = torch.zeros((rows * world, W, 4))
                dist.all_gather_into_tensor(hr, rgba_part.cpu())
                rgba_all.copy_(hr)
            else:
                dist.all_gather_into_tensor(rgba_all, rgba_part)
            return
        if rank == 0:
            ds.resolve_device(film.data_ptr(), H * W, rgb_dev.data_ptr(), alpha_dev.data_ptr())      # ImageFilm::WriteImage normalisation, on the device

    def handover():
        """rank 0: the resolved film into the page-locked host buffers (reused every frame)"""
        if rank != 0:
            return None
        if merge == "reduce_scatter":
            host_rgba_t.copy_(rgba_all[:H], non_blocking=True)
        else:
            host_rgb_t.copy_(rgb_dev, non_blocking=True); host_alpha_t.copy_(alpha_dev, non_blocking=True)
        torch.cuda.synchronize()
        return host_rgb, host_alpha

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # one counted frame: ray / node / triangle counts of this rank's shard (deterministic per frame)
    ds.set_counting(True); ds.reset_counters()
    step(); fence()
    cnt = ds.counters()
    ds.set_counting(False)
    for _ in range(warmup):
        step()
    fence()
    stats = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        stats.append(ds.last_stats())       # HIP events around the kernels on their stream
    fence()
    elapsed = time.perf_counter() - t0
    # the same frames with the hand-over to host memory at the end of each (PCIe): reported beside the device-resident figure, never as `value`
    t1 = time.perf_counter()
    for _ in range(steps):
        step(); handover()
    fence()
    elapsed_host = time.perf_counter() - t1
    rays_local = cnt["closest_rays"] + cnt["any_rays"]
    if os.environ.get("PBRT_BENCH_COUNTERS"):
        print("COUNTERS " + json.dumps({k: int(v) for k, v in cnt.items()}), file=sys.stderr, flush=True)
    # the traversal kernels of the frame: rt::render_kernel, or the rt::pipe_trace_kernel (+ rt::pipe_march_kernel, which traces the marches' shadow rays) launches summed
    k_ms_local = float(np.mean([st["trace_ms"] + st.get("march_ms", 0.0) for st in stats]))
    render_ms_local = float(np.mean([st["render_ms"] for st in stats]))
    tot = torch.tensor([float(rays_local), float(cnt["camera_rays"])], dtype=torch.float64, device="cuda")
    tmax = torch.tensor([elapsed, elapsed_host], dtype=torch.float64, device="cuda")
    per_rank = torch.zeros(world * 3, dtype=torch.float64, device="cuda")
    per_rank[3 * rank] = k_ms_local; per_rank[3 * rank + 1] = render_ms_local; per_rank[3 * rank + 2] = float(rays_local)
    if dist is not None:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    rays_total, cam_total = float(tot[0].item()), float(tot[1].item())
    elapsed = float(tmax[0].item()); elapsed_host = float(tmax[1].item())
    out = None
    if rank == 0:
        if dump_film:
            np.savez(dump_film, rgb=host_rgb, alpha=host_alpha)
        ms_per_step = elapsed / steps * 1e3
        # the dominant kernel: rt::render_kernel (megakernel) or the rt::pipe_trace_kernel launches of the queue pipeline
        k_ms = k_ms_local
        pipeline = bool(stats[-1]["pipeline"])
        alg_bytes = 8 * cnt["nodes_visited"] + 4 * cnt["leaf_refs"] + 48 * cnt["tri_tests"] + 48 * rays_local
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        pr = per_rank.cpu().numpy().reshape(world, 3)
        out = {
            "metric": "Mrays/s (primary+secondary: every Scene::Intersect + Scene::IntersectP)",
            "value": round(rays_total * steps / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            # 2 (since round 4): the timed step ends with the RESOLVED film in device memory; 1 (rounds 1-3): it ended with the film in page-locked host memory --
            # that figure is still printed, as host_handover.value, for like-for-like comparisons across the change
            "metric_version": 2,
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_per_step, 3),
            "s_per_frame": round(ms_per_step / 1e3, 5),
            "host_handover": {"ms_per_step": round(elapsed_host / steps * 1e3, 3), "value": round(rays_total * steps / elapsed_host / 1e6, 3),
                              "note": "the same steps with the resolved film copied to page-locked host memory at the end of each (rank 0, %d x %d x 4 floats over PCIe); "
                                      "`value` and `ms_per_step` end with the resolved film in device memory" % (W, H)},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": label, "camera_samples_per_frame": int(cam_total), "rays_per_frame": int(rays_total),
                       "rays_per_camera_sample": round(rays_total / max(cam_total, 1), 3),
                       "kd_nodes": int(info.n_nodes), "kd_build_s": round(info.build_seconds, 4),
                       "parallelism": "%s dealt round-robin to %d rank(s); %s; accelerator built once per node (%.2f s scene create on rank 0)"
                                      % ("2-D tiles of %dx%d pixels" % tuple(tiles_used) if args.tile_2d > 0 else "tiles of %d consecutive pixels" % args.tile_pixels, world,
                                         {"allreduce": "RCCL all-reduce(sum) of the 5-plane film, rank 0 resolves", "reduce_scatter": "one RCCL reduce-scatter of the film (part r = the 5 planes of rank r's rows), per-rank resolve, one all-gather of the resolved RGBA rows",
                                          "none": "single rank"}[merge], t_create),
                       "rng": "counter-based keyed RNG, seed 0"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 5), "traffic": None,
                         "kernel": (("rt::pipe_trace_kernel<COUNT=false,...> + rt::pipe_march_kernel<COUNT=false,...> (the traversal kernels of the queue pipeline with a medium; all %d launch pairs of a frame summed: trace %.1f ms, march %.1f ms)"
                                     % (stats[-1]["iterations"], float(np.mean([st["trace_ms"] for st in stats])), float(np.mean([st.get("march_ms", 0.0) for st in stats]))))
                                    if stats[-1].get("march_ms", 0.0) > 0 else
                                    "rt::pipe_trace_kernel<COUNT=false,...> (persistent trace waves of the queue pipeline; all %d launches of a frame summed)" % stats[-1]["iterations"])
                                   if pipeline else "rt::render_kernel<COUNT=false,...> (persistent megakernel)",
                         "kernel_ms": round(k_ms, 3),
                         # the same bytes over ALL kernels of the frame's render part (megakernel: the same kernel; pipeline: shade passes + trace
                         # launches), and over the whole step (film gather, resolve, host hand-over included)
                         "frac_frame_kernels": round(alg_bytes / (render_ms_local * 1e-3) / 1e9 / 8000.0, 5),
                         "frac_step": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / 8000.0, 5),
                         "frame_kernels_ms": {"render": round(render_ms_local, 3), "shade_launches": round(float(np.mean([st.get("shade_ms", 0.0) for st in stats])), 3),
                                              "film_gather": round(float(np.mean([st["gather_ms"] for st in stats])), 3)},
                         "pipeline_iterations": int(stats[-1]["iterations"]), "pipeline_slots": int(stats[-1]["slots"]),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "bytes_per_ray": round(alg_bytes / max(rays_local, 1), 1),
                         "nodes_per_ray": round(cnt["nodes_visited"] / max(rays_local, 1), 2),
                         "tri_tests_per_ray": round(cnt["tri_tests"] / max(rays_local, 1), 2),
                         # the ceiling measured for this access pattern (tools/gather_bench.hip, profiles/r02_gather_bench.txt): the chip
                         # sustains ~56 G dependent gathers/s that miss L2, whatever their width; one gather per node visit, leaf list and
                         # triangle record is the worst case the layout allows
                         "gather_ceiling": {"gathers_per_launch": int(cnt["nodes_visited"] + cnt["tri_tests"] + cnt["leaf_refs"] // 2),
                                            "l2_miss_gathers_per_s_peak": 56e9,
                                            "ms_if_every_gather_missed_l2": round((cnt["nodes_visited"] + cnt["tri_tests"] + cnt["leaf_refs"] // 2) / 56e9 * 1e3, 2)}},
            "per_rank": [{"rank": r, "kernel_ms": round(float(pr[r, 0]), 3), "render_ms": round(float(pr[r, 1]), 3), "rays": int(pr[r, 2])} for r in range(world)],
        }
        # HBM traffic of the same kernel from the PMC passes of tools/profile_r.sh (separate rocprofv3 runs of this very
        # command; counters and corrections as MI355X_MICROARCH.md prescribes), committed under profiles/
        if world == 1:
            attach_profile(out, name, alg_bytes, pkg.code_id())
        if world == 1 and with_cpu and name.startswith("c4full"):
            # the reference's own kd build of 10 M triangles takes tens of minutes on one core and its 2048 x 2048 @ 256 frame days: no CPU leg inside a bench run
            why = "the reference's single-threaded kd build of 10 M triangles alone exceeds the bench run's budget (1 M: ~90 s, superlinear); see the 1 M-triangle C4 sub-record for the same frame's CPU leg"
            out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 1, "kind": "reference", "reason": why}
            out["parity"] = {"value": None, "reason": why + "; the 245.7 M-node tree is traced bit-exactly against the oracle in tests/test_gpu_c4_full.py"}
        elif world == 1 and with_cpu:
            out["cpu_baseline"] = cpu_baseline(pkg, name, crop)
            if out["cpu_baseline"].get("value"):
                out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        else:
            out["cpu_baseline"] = None
    ds.close()
    del film
    if rank == 0 and world == 1 and with_cpu and "parity" not in out:
        torch.cuda.empty_cache()
        try:
            out["parity"] = parity_leg(pkg, name, crop, device_index)
        except Exception as e:                               # the line must not be lost to its third leg
            out["parity"] = {"value": None, "reason": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    # the headline is the LARGEST stated single-GPU configuration of BASELINE.json: configs[2] (C3: 1 M-triangle soup, DirectLighting,
    # 1920x1080 @ 16 spp).  configs[1] (C2, Cornell) is cache-resident and VALU-bound; it is reported as a sub-record.
    ap.add_argument("--workload", default=HEADLINE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # the other BASELINE.json configs at the size one GPU holds, reported as full sub-records in "workloads" (N = 1 only)
    ap.add_argument("--extra-workloads", default="c2,p1000000,c4,c5,c3_tuned,p1000000_tuned,c4full")
    ap.add_argument("--no-extra", action="store_true")
    # N > 1: the configurations BASELINE.json states for several GPUs, as sub-records with `per_rank` next to the headline: "auto" = C4 as stated
    # (10 M triangles, 2048 x 2048 @ 256 spp, "tiles over 8 x MI355X") at 8 ranks, C5 ("4 MI355X") at 4 and 8 ranks
    ap.add_argument("--multi-workloads", default="auto")
    ap.add_argument("--dump-film", default=None, help="rank 0 writes the last resolved film of the headline workload here (.npz)")
    # 48: with the 1028-pixel sample rows of the default frame, 64-pixel tiles give 16.06 tiles per row, so one rank owns the
    # same columns for ~16 consecutive rows and whole 16x16 film-gather blocks fall to a single rank (measured at 8 ranks:
    # rank share 10.06 ms with 64, 9.61 ms with 48)
    ap.add_argument("--tile-pixels", type=int, default=48)
    # 2-D tiles of T x T pixels (multiples of the film gather's 16x16 blocks): a rank's samples fall on compact pieces of the film, so its
    # gather touches the blocks around them only; 0 = the 1-D tiles above
    ap.add_argument("--tile-2d", type=int, default=64)
    ap.add_argument("--tile-h", type=int, default=0, help="height of the 2-D tiles when it differs from --tile-2d (experiments)")
    ap.add_argument("--merge", choices=["allreduce", "reduce_scatter"], default="reduce_scatter")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even for one rank: exercises the N > 1 code path on a single GPU")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC only on these hosts: without it RCCL's buffer exchange fails with hipIpcGetMemHandle errors
    import torch
    import __graft_entry__ as entry
    pkg = entry.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # tests/test_multirank_gpu.py runs two ranks on ONE GPU over gloo (films staged through the host by the backend) so that the
    # N > 1 branch of this file executes before an 8-GPU node is available; the driver's runs use one GPU per rank over RCCL
    backend = os.environ.get("PBRT_BENCH_BACKEND", "nccl")
    device_index = 0 if os.environ.get("PBRT_BENCH_SAME_GPU") else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1 or args.force_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if torch.cuda.device_count() <= device_index:
            raise SystemExit("bench.py: rank %d wants GPU %d but only %d are visible (one process per GPU: launch with --nproc-per-node <= #GPUs)"
                             % (rank, device_index, torch.cuda.device_count()))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index),
                                        timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
            probe = torch.ones(8, device="cuda")                     # first collective: communicator set-up errors surface here, not mid-frame
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe[0].item()) != world:
                raise RuntimeError("all_reduce probe returned %r for world size %d" % (probe[0].item(), world))
        except Exception as e:
            raise SystemExit("bench.py: torch.distributed (%s) set-up failed on rank %d / %d, MASTER %s:%s: %s: %s" % (
                backend, rank, world, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"), type(e).__name__, e))

    out = run_workload(args.workload, args, pkg, torch, dist, world, rank, device_index, args.steps, args.warmup,
                       with_cpu=not args.no_cpu_baseline, dump_film=args.dump_film)
    if world > 1:
        multi = ({8: "c4full,c5", 4: "c5"}.get(world, "") if args.workload == HEADLINE else "") if args.multi_workloads == "auto" else args.multi_workloads
        extras = [] if args.no_extra else [w for w in multi.split(",") if w]
    else:
        extras = [] if (args.no_extra or args.workload != HEADLINE) else [w for w in args.extra_workloads.split(",") if w]
    records = []
    for w in extras:
        # a sub-record must never cost the headline: whatever goes wrong in an extra workload (memory, a collective timing out) is recorded in its place
        rec, err = None, None
        try:
            rec = run_workload(w, args, pkg, torch, dist, world, rank, device_index, min(args.steps, 2 if w == "c4full" else 3), 1, with_cpu=not args.no_cpu_baseline,
                               dump_film=args.dump_film.replace(".npz", "_%s.npz" % w) if args.dump_film else None)
        except BaseException as e:                                       # incl. SystemExit raised by run_workload
            err = "%s: %s" % (type(e).__name__, e)
        if dist is not None:                                             # all ranks agree on whether to go on
            try:
                flag = torch.tensor([0.0 if err is None else 1.0], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if flag.item() > 0 and err is None:
                    err = "another rank failed in this workload"
            except BaseException as e:
                err = (err or "") + " | agreement all-reduce failed: %s" % e
        if err is not None:
            records.append({"workload": w, "error": err[:500]})
            break
        if rec is not None:
            records.append({k: rec[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "steps", "host_handover", "config", "roofline", "per_rank", "cpu_baseline", "parity") if k in rec} |
                           ({"speedup_vs_cpu_baseline": rec["speedup_vs_cpu_baseline"]} if "speedup_vs_cpu_baseline" in rec else {}) | {"workload": w})
    if dist is not None:                                                 # before the line is printed: RCCL writes its version banner to stdout when the group goes down
        try:
            dist.barrier()
            dist.destroy_process_group()
        except BaseException:
            pass
    try:                                                                 # RCCL's banner sits in the C library's stdout buffer until exit: out with it, so that the JSON line is the LAST line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except BaseException:
        pass
    if rank == 0:
        if records:
            out["workloads"] = records
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
