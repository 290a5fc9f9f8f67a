"""GPU box: the by-vertex path pipeline (rt_pipe_vertex.h) against the megakernel: films bit-identical, counters identical."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
from pbrt_v1_amd import scenes

CASES = [
    dict(xres=96, yres=96, integrator="path", maxdepth=5, xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell"),
    dict(xres=128, yres=128, integrator="path", maxdepth=8, xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell", soup_tris=30000, soup_materials=True),
    dict(xres=96, yres=96, integrator="path", maxdepth=8, sampler="lowdiscrepancy", pixelsamples=4, soup_tris=50000, soup_materials=True),
    dict(xres=64, yres=64, integrator="path", maxdepth=3, sampler="random", xsamples=2, ysamples=2, soup_tris=2000),
    dict(xres=64, yres=64, integrator="path", maxdepth=0, xsamples=1, ysamples=1, soup_tris=500),
    dict(xres=96, yres=96, integrator="directlighting", xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell", soup_tris=20000,
         volume_integrator='"single" "float stepsize" [40]', world_kwargs=dict(volume='"float g" [.2]')),
    dict(xres=80, yres=80, integrator="path", maxdepth=5, xsamples=2, ysamples=2, jitter=True, soup_tris=20000, soup_materials=True,
         volume_integrator='"single" "float stepsize" [60]', world_kwargs=dict(volume='"float g" [-.1] "color Le" [.001 .001 .002]')),
    dict(xres=96, yres=96, integrator="whitted", xsamples=2, ysamples=1, soup_tris=25000, soup_materials=True),
]
ok = True
for cfg in CASES:
    ps = pkg.ParsedScene(text=scenes.cornell_scene(keyed=True, **cfg))
    assert ps.valid and ps.errors == 0
    ds = pkg.DeviceScene(ps)
    ds.render()
    res = {}
    for tag, env, counting in (("mega_count", dict(PBRT_HIP_PIPELINE="0"), True), ("pipe_ray_count", dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_PIPE_VERTEX="0"), True),
                               ("pipe_vertex_count", dict(PBRT_HIP_PIPELINE="1"), True), ("pipe_vertex", dict(PBRT_HIP_PIPELINE="1"), False),
                               ("pipe_vertex_small", dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_PIPE_SLOTS="1024"), False), ("mega", dict(PBRT_HIP_PIPELINE="0"), False),
                               ("pipe_vertex_overlap", dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_OVERLAP="1"), False), ("pipe_vertex_overlap_small", dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_OVERLAP="1", PBRT_HIP_PIPE_SLOTS="2048"), True),
                               ("pipe_ray_overlap", dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_OVERLAP="1", PBRT_HIP_PIPE_VERTEX="0"), True), ("pipe_ray_overlap_small", dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_OVERLAP="1", PBRT_HIP_PIPE_VERTEX="0", PBRT_HIP_PIPE_SLOTS="1024"), False)):
        for k in ("PBRT_HIP_PIPELINE", "PBRT_HIP_PIPE_VERTEX", "PBRT_HIP_PIPE_SLOTS", "PBRT_HIP_OVERLAP"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ds.set_counting(counting); ds.reset_counters(); ds.clear_film(); ds.render()
        res[tag] = (ds.film_accum().copy(), ds.counters() if counting else None, ds.last_stats())
    ref_acc, ref_cnt, _ = res["mega_count"]
    for tag, (acc, cnt, st) in res.items():
        same = np.array_equal(acc, ref_acc)
        csame = cnt is None or all(cnt[k] == ref_cnt[k] for k in ref_cnt if k != "stack_overflows")
        print(cfg.get("soup_tris", 0), cfg.get("sampler", "stratified"), tag, "film_equal", same, "counters_equal", csame, "iters", st["iterations"], "maxdiff", float(np.abs(acc - ref_acc).max()))
        if not csame:
            print("   ", cnt, ref_cnt)
        ok = ok and same and csame
    ds.close()
print("ALL OK" if ok else "MISMATCH")
