"""Exploratory parity run (GPU box): HIP path vs the compiled reference (oracle/_ref) on a config matrix.
Prints one line per config.  Not a test -- tests/ hold the asserted versions."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
REF = g.load_ref_runner()
from pbrt_v1_amd import scenes

def compare(name, text, keyed=True):
    t0 = time.time()
    rgb, alpha, cnt, ms = pkg.render_text(text)
    t1 = time.time()
    ref_rgb, ref_alpha, st = REF.run_reference(text, keyed=keyed)
    t2 = time.time()
    d = rgb - ref_rgb
    l2 = np.sqrt((d ** 2).sum(-1))
    out = dict(cfg=name, gpu_ms=round(ms, 3), wall_gpu=round(t1 - t0, 2), ref_render_s=round(st["render_s"], 3),
               maxabs=float(np.abs(d).max()), rmse=float(np.sqrt((d ** 2).mean())), mean_l2=float(l2.mean()),
               frac_lt_1e4=float((l2 < 1e-4).mean()), alpha_max=float(np.abs(alpha - ref_alpha).max()),
               closest=(cnt["closest_rays"], st["closest_rays"]), any=(cnt["any_rays"], st["any_rays"]),
               mean=(float(rgb.mean()), float(ref_rgb.mean())), bad=cnt["bad_samples"], spills=cnt["stack_overflows"],
               nodes=cnt["nodes_visited"], tris=cnt["tri_tests"])
    print(json.dumps(out), flush=True)
    return out

cfgs = []
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    cfgs = [("whitted_pt_64", dict(xres=64, yres=64, integrator="whitted", world_kwargs=dict(point_light=True, area_light=False)))]
else:
    cfgs = [
        ("whitted_point_128", dict(xres=128, yres=128, integrator="whitted", world_kwargs=dict(point_light=True, area_light=False))),
        ("whitted_area_128", dict(xres=128, yres=128, integrator="whitted")),
        ("direct_all_128", dict(xres=128, yres=128, integrator="directlighting")),
        ("direct_all_ns4_64", dict(xres=64, yres=64, integrator="directlighting", world_kwargs=dict(light_nsamples=4))),
        ("direct_one_128", dict(xres=128, yres=128, integrator="directlighting", integrator_params='"string strategy" ["one"]')),
        ("path_128_4spp", dict(xres=128, yres=128, integrator="path", xsamples=2, ysamples=2)),
        ("path_jitter_64_4spp", dict(xres=64, yres=64, integrator="path", xsamples=2, ysamples=2, jitter=True)),
        ("path_mitchell_64", dict(xres=64, yres=64, integrator="path", xsamples=2, ysamples=2, pixel_filter="mitchell")),
        ("direct_soup5k_128", dict(xres=128, yres=128, integrator="directlighting", soup_tris=5000)),
        ("path_soup5k_128_4", dict(xres=128, yres=128, integrator="path", xsamples=2, ysamples=2, soup_tris=5000)),
        ("whitted_mirror_128", dict(xres=128, yres=128, integrator="whitted", world_kwargs=dict(mirror_quad=True))),
        ("whitted_glass_128", dict(xres=128, yres=128, integrator="whitted", world_kwargs=dict(glass_sphere_tris=scenes.icosphere((200, 120, 250), 90, 2)))),
        ("path_glass_mirror_64_4", dict(xres=64, yres=64, integrator="path", xsamples=2, ysamples=2, world_kwargs=dict(mirror_quad=True, glass_sphere_tris=scenes.icosphere((200, 120, 250), 90, 2)))),
        ("c1_whitted_512", dict(xres=512, yres=512, integrator="whitted")),
    ]
for name, kw in cfgs:
    try:
        compare(name, scenes.cornell_scene(keyed=True, count=True, **kw))
    except Exception as e:
        print(json.dumps(dict(cfg=name, error=repr(e))), flush=True)
