"""Round 4: what a DirectLighting "weighted" frame costs next to "one" / "all" on the same scene (1x MI355X).  Cornell + 1 M-triangle soup with the
material mix (glass / mirror recursion), four delta lights, 1024 x 1024 @ 4 spp.  Prints one line per strategy; for "weighted" also the five parts
(count pass, scan, survey pass, recurrence kernel, frame pass) and the frame's shading points.  -> profiles/r04_weighted_timing.txt"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); from pbrt_v1_amd import scenes
SPOT = 'LightSource "spot" "point from" [278 540 100] "point to" [200 0 330] "color I" [600000 500000 400000] "float coneangle" [35] "float conedeltaangle" [12]\n'
DISTANT = 'LightSource "distant" "point from" [0.3 1 -0.8] "point to" [0 0 0] "color L" [1.5 1.6 2.0]\n'
POINT2 = 'LightSource "point" "point from" [100 300 100] "color I" [150000 250000 150000]\n'
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_more = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # extra point lights (how the recurrence's cost per point grows with the number of lights)
if n_more:
    rng = np.random.default_rng(5)
    POINT2 += "".join('LightSource "point" "point from" [%.1f %.1f %.1f] "color I" [%.0f %.0f %.0f]\n' % (*rng.uniform((40, 60, 40), (510, 520, 500)), *rng.uniform(20000, 160000, 3))
                      for _ in range(n_more))
for strat in ("one", "all", "weighted"):
    text = scenes.cornell_scene(xres=1024, yres=1024, xsamples=2, ysamples=2, jitter=True, keyed=True, integrator="directlighting",
                                integrator_params='"string strategy" ["%s"]' % strat, soup_tris=n_tris, soup_materials=True,
                                world_kwargs=dict(point_light=True, area_light=False, extra=SPOT + DISTANT + POINT2))
    ps = pkg.ParsedScene(text=text); del text
    ds = pkg.DeviceScene(ps)
    ds.set_counting(True); ds.render(); cnt = ds.counters()
    ds.set_counting(False)
    ms = []
    for i in range(4):
        ds.clear_film(); ds.render(); st = ds.last_stats()
        if i: ms.append(st["render_ms"])
    line = "%-8s render %.2f ms  rays %d (closest %d, shadow %d)  %.0f Mrays/s" % (strat, np.mean(ms), cnt["closest_rays"] + cnt["any_rays"], cnt["closest_rays"], cnt["any_rays"],
                                                                              (cnt["closest_rays"] + cnt["any_rays"]) / np.mean(ms) / 1e3)
    if strat == "weighted":
        w = st["weighted_ms"]
        line += "  | shading points %d: count %.2f, scan %.2f, survey %.2f, recurrence %.2f (%.0f ns/point), frame %.2f ms" % (
            st["weighted_points"], w[0], w[1], w[2], w[3], 1e6 * w[3] / max(1, st["weighted_points"]), w[4])
    print(line, flush=True)
    ds.close()
