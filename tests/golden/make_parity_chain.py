"""Fixtures that harden the parity chain within what this image allows (VERDICT r01 "next round" item 6).  Authoring container only:

    python tests/golden/make_parity_chain.py

tests/golden/chain/
  t0_*.npz       scenes whose image does not depend on any random draw, rendered by the RNG-UNTOUCHED reference (oracle/_ref/pbrt_ref:
                 its own MT19937, its own samplers, no helper plugin in the scene) -- and, asserted here, bit-identical to the keyed
                 build's film.  The device / oracle are compared with these, so the keyed RNG link-time override is not in the loop.
  api_*.npz      the Cornell box of configs[0] / configs[1] issued to the reference through hand-written pbrt* API calls
                 (ref_driver.cpp BuiltinCornell): no scene text and no tokenizer in the reference run.  Asserted here: the same film
                 as the reference run of the equivalent scene FILE (which goes through the repo's scene_parser.h).
  t2_cornell.npz converged Cornell path-traced image (64x64 @ 1024 spp) from the reference's native MT19937 stream, twice with
                 different sample patterns: the second tells how far two correct renderers are apart.
  bad_*.npz      emitters that make Scene::Render's sanity check fire (scene.cpp:60-74: NaN, negative, infinite luminance).
  sinc_*.npz     the windowed-sinc pixel filter (filters/sinc.cpp:41-53)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from pbrt_v1_amd import scenes
REF = g.load_ref_runner()
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chain")
os.makedirs(OUT, exist_ok=True)
blob = scenes.icosphere((200, 120, 250), 90, 1)


def save(name, text, rgb, alpha, st, **extra):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), scene=np.array(text), rgb=rgb, alpha=alpha,
                        stats=np.array(json.dumps({k: v for k, v in st.items() if k != "stats"})), **extra)
    print(name, rgb.shape, "mean %.5f" % float(rgb.mean()), {k: extra[k] for k in extra if np.ndim(extra[k]) == 0})


SPOT = 'LightSource "spot" "point from" [278 540 100] "point to" [200 0 330] "color I" [600000 500000 400000] "float coneangle" [35] "float conedeltaangle" [12]\n'
DISTANT = 'LightSource "distant" "point from" [0.3 1 -0.8] "point to" [0 0 0] "color L" [1.5 1.6 2.0]\n'
SPHERE = 'AttributeBegin\nMaterial "mirror"\nTranslate 300 330 420\nRotate -70 1 0.2 0\nShape "sphere" "float radius" [90] "float zmin" [-60] "float zmax" [70] "float phimax" [250]\nAttributeEnd\n'
MESH = "AttributeBegin\nMaterial \"matte\" \"color Kd\" [.7 .6 .3]\nTranslate 160 110 330\n%sAttributeEnd\n" % scenes.smooth_mesh_text(radius=100, with_n=True, with_uv=True)
T0 = {   # point / spot / distant lights only, unjittered strata, Whitted or DirectLighting: no draw reaches the image
    "t0_whitted_point": dict(xres=48, yres=48, integrator="whitted", world_kwargs=dict(point_light=True, area_light=False)),
    "t0_direct_spot_distant": dict(xres=40, yres=40, integrator="directlighting", xsamples=2, ysamples=2, pixel_filter="mitchell",
                                   world_kwargs=dict(area_light=False, extra=SPOT + DISTANT)),
    "t0_whitted_glass_mirror": dict(xres=48, yres=48, integrator="whitted", maxdepth=4,
                                    world_kwargs=dict(point_light=True, area_light=False, mirror_quad=True, glass_sphere_tris=blob)),
    "t0_whitted_sphere_mesh_soup": dict(xres=40, yres=40, integrator="whitted", soup_tris=800, pixel_filter="gaussian",
                                        world_kwargs=dict(point_light=True, area_light=False, extra=SPHERE + MESH)),
    "t0_direct_grid": dict(xres=32, yres=32, integrator="directlighting", accelerator="grid", soup_tris=500, world_kwargs=dict(point_light=True, area_light=False)),
}


def main():
    only = set(sys.argv[1:])
    for name, kw in ({} if only else T0).items():
        text = scenes.cornell_scene(**kw)                                  # plain plugin names: stratified sampler, kdtree, MT19937
        rgb, alpha, st = REF.run_reference(text, keyed=False)
        rk, ak, _ = REF.run_reference(scenes.cornell_scene(keyed=True, count=True, **kw), keyed=True)
        assert np.array_equal(rgb, rk) and np.array_equal(alpha, ak), name      # the keyed build renders the same film when no draw matters
        save(name, text, rgb, alpha, st)
    # ---- parser common mode: hand-written API calls vs the scene file
    for kind, kw in () if only else (("c1", dict(integrator="whitted", xsamples=1, ysamples=1, jitter=False, pixel_filter="box")),
                     ("c2", dict(integrator="path", maxdepth=5, xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell"))):
        text = scenes.cornell_scene(xres=64, yres=64, keyed=True, count=True, **kw)
        rgb_file, alpha_file, st = REF.run_reference(text, keyed=True)
        rgb_api, alpha_api, st_api = REF.run_reference(("builtin", kind, 64), keyed=True)
        assert np.array_equal(rgb_file, rgb_api) and np.array_equal(alpha_file, alpha_api), kind
        assert st["closest_rays"] == st_api["closest_rays"] and st["any_rays"] == st_api["any_rays"]
        save("api_cornell_" + kind, text, rgb_api, alpha_api, st_api)
    # ---- T2: two converged renders by the native MT19937 stream
    kw = dict(xres=64, yres=64, integrator="path", maxdepth=5, pixel_filter="box")
    if only: return rest()
    text_a = scenes.cornell_scene(xsamples=32, ysamples=32, jitter=True, **kw)
    text_b = scenes.cornell_scene(sampler="random", xsamples=32, ysamples=32, **kw)
    rgb_a, alpha_a, st = REF.run_reference(text_a, keyed=False)
    rgb_b, _, _ = REF.run_reference(text_b, keyed=False)
    save("t2_cornell", text_a, rgb_a, alpha_a, st, rgb_second=rgb_b, rmse_between=np.float64(np.sqrt(((rgb_a - rgb_b) ** 2).mean())))
    rest()


def rest():
    # ---- scene.cpp:60-74: an infinite emitter ("1e39" overflows float, as in the reference's atof) and its negative twin
    pos = 'AreaLightSource "area" "color L" [1e39 1e39 1e39]\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [200 548 200 350 548 200 280 548 350]\n'
    neg = 'AreaLightSource "area" "color L" [-1e39 -1e39 -1e39]\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [100 548 300 250 548 300 180 548 450]\n'
    small_neg = 'AreaLightSource "area" "color L" [-4 -5 -6]\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [300 548 300 450 548 300 380 548 450]\n'
    for name, kw, extra in (("bad_inf_nan_direct", dict(integrator="directlighting", xsamples=2, ysamples=1, jitter=True), pos + neg),
                            ("bad_negative_path", dict(integrator="path", xsamples=2, ysamples=2, jitter=True), small_neg)):
        text = scenes.cornell_scene(xres=32, yres=32, keyed=True, count=True, world_kwargs=dict(extra=extra), **kw)
        rgb, alpha, st = REF.run_reference(text, keyed=True)
        assert st["radiance_warnings"] > 0 and np.isfinite(rgb).all(), (name, st["radiance_warnings"])
        save(name, text, rgb, alpha, st, radiance_warnings=np.int64(st["radiance_warnings"]))
    for name, kw in (("sinc_whitted", dict(integrator="whitted", xsamples=2, ysamples=2, jitter=True, pixel_filter="sinc")),
                     ("sinc_tau_path", dict(integrator="path", xsamples=2, ysamples=2, jitter=True, pixel_filter="sinc",
                                            filter_params='"float xwidth" [3] "float ywidth" [2.5] "float tau" [2]'))):
        text = scenes.cornell_scene(xres=36, yres=36, keyed=True, count=True, **kw)
        rgb, alpha, st = REF.run_reference(text, keyed=True)
        save(name, text, rgb, alpha, st)


if __name__ == "__main__":
    main()
