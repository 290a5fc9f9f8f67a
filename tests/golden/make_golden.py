"""Generate the golden fixtures in tests/golden/ by running the UNMODIFIED reference (oracle/_ref/pbrt_ref_keyed,
built from /root/reference by oracle/ref/Makefile) on small scenes.  Runs only in the authoring container.

    python tests/golden/make_golden.py

Each fixture <name>.npz holds: the scene text, the reference's float film (rgb, alpha: the arguments of
WriteRGBAImage, i.e. before half quantisation), its ray counts (countaccel plugin) and the kd-tree / triangle-test
counters printed by StatsPrint.  probe_*.npz hold per-camera-sample records from the probe integrator plugin
(oracle/ref/probe_integrator.cpp): camera ray, closest hit (t, p, n, u, v), shadow-segment occlusion.
Fixtures are DATA (inputs + expected outputs); no reference source text is stored."""
import json, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from pbrt_v1_amd import scenes
import importlib.util
_spec = importlib.util.spec_from_file_location("pbrt_ref_runner", os.path.join(ROOT, "oracle", "ref_runner.py"))
REF = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(REF)

OUT = os.path.dirname(os.path.abspath(__file__))
blob = scenes.icosphere((200, 120, 250), 90, 1)

SPOT = 'LightSource "spot" "point from" [278 540 100] "point to" [200 0 330] "color I" [600000 500000 400000] "float coneangle" [35] "float conedeltaangle" [12]\n'
DISTANT = 'LightSource "distant" "point from" [0.3 1 -0.8] "point to" [0 0 0] "color L" [1.5 1.6 2.0]\n'
SPOT_XF = ('AttributeBegin\nTranslate 30 0 10\nRotate 20 0 1 0\nLightSource "spot" "point from" [250 500 200] "point to" [300 0 300] "color I" [300000 300000 500000]\n'
           'AttributeEnd\n')
DISTANT_XF = 'AttributeBegin\nRotate -35 1 0 0.2\nScale 2 2 2\nLightSource "distant" "point from" [0 1 0] "point to" [0 0 0] "color L" [2 2 1.5]\nAttributeEnd\n'

SPHERES = ('AttributeBegin\nMaterial "matte" "color Kd" [.7 .6 .2]\nTranslate 150 112 350\nShape "sphere" "float radius" [100]\nAttributeEnd\n'
           'AttributeBegin\nMaterial "glass" "float index" [1.5]\nTranslate 400 120 200\nRotate 30 1 0 0\nScale 1 1.3 1\nShape "sphere" "float radius" [80]\nAttributeEnd\n'
           'AttributeBegin\nMaterial "mirror"\nTranslate 300 330 420\nRotate -70 1 0.2 0\nShape "sphere" "float radius" [90] "float zmin" [-60] "float zmax" [70] "float phimax" [250]\nAttributeEnd\n'
           'AttributeBegin\nMaterial "plastic" "color Kd" [.2 .3 .7] "float roughness" [.15]\nReverseOrientation\nTranslate 120 380 250\nScale -1 1 1\nShape "sphere" "float radius" [60]\nAttributeEnd\n')

QUADRICS = ('AttributeBegin\nMaterial "matte" "color Kd" [.7 .6 .2]\nTranslate 150 0 330\nRotate -90 1 0 0\nShape "cylinder" "float radius" [70] "float zmin" [0] "float zmax" [180]\n'
            'Translate 0 0 180\nShape "disk" "float radius" [70]\nAttributeEnd\n'
            'AttributeBegin\nMaterial "mirror"\nTranslate 400 200 380\nRotate 35 0 1 0\nRotate 60 1 0 0\nShape "disk" "float radius" [110] "float innerradius" [40] "float phimax" [300] "float height" [10]\nAttributeEnd\n'
            'AttributeBegin\nMaterial "glass" "float index" [1.4]\nTranslate 380 90 180\nRotate 20 0 0 1\nScale 1 .8 1\nShape "cylinder" "float radius" [60] "float zmin" [-50] "float zmax" [70] "float phimax" [270]\nAttributeEnd\n'
            'AttributeBegin\nMaterial "plastic" "color Kd" [.2 .3 .7]\nReverseOrientation\nTranslate 100 400 250\nRotate 45 1 1 0\nShape "cylinder" "float radius" [40] "float zmin" [-60] "float zmax" [60]\nAttributeEnd\n')

QLIGHTS = ('AttributeBegin\nAreaLightSource "area" "color L" [12 10 6]\nMaterial "matte" "color Kd" [0 0 0]\nTranslate 150 300 300\nShape "sphere" "float radius" [40]\nAttributeEnd\n'
           'AttributeBegin\nAreaLightSource "area" "color L" [6 9 14]\nMaterial "matte" "color Kd" [.2 .2 .2]\nTranslate 420 400 250\nRotate 70 1 0 0.3\nShape "disk" "float radius" [60]\nAttributeEnd\n'
           'AttributeBegin\nAreaLightSource "area" "color L" [9 4 4]\nMaterial "matte" "color Kd" [.1 .1 .1]\nReverseOrientation\nTranslate 300 60 150\nRotate 90 0 1 0\nScale 1 1.2 1\nShape "cylinder" "float radius" [25] "float zmin" [-80] "float zmax" [80]\nAttributeEnd\n')

QUADRICS2 = ('AttributeBegin\nMaterial "matte" "color Kd" [.7 .6 .2]\nTranslate 140 0 330\nRotate -90 1 0 0\nShape "cone" "float radius" [80] "float height" [200]\nAttributeEnd\n'
             'AttributeBegin\nMaterial "mirror"\nTranslate 400 60 380\nRotate -70 1 0 0.2\nShape "paraboloid" "float radius" [90] "float zmin" [20] "float zmax" [150] "float phimax" [300]\nAttributeEnd\n'
             'AttributeBegin\nMaterial "plastic" "color Kd" [.2 .3 .7]\nTranslate 300 330 200\nRotate 40 1 0 1\nScale 60 60 60\nShape "hyperboloid" "point p1" [1 0 -1.2] "point p2" [.8 .9 1.1]\nAttributeEnd\n'
             'AttributeBegin\nMaterial "glass" "float index" [1.3]\nReverseOrientation\nTranslate 120 300 160\nRotate 100 0 1 0\nShape "hyperboloid" "point p1" [40 0 0] "point p2" [20 30 70] "float phimax" [270]\nAttributeEnd\n')

# a second emitter of two triangles next to the Cornell ceiling light (each a ShapeSet: one RandomFloat() per light sample) and a second point light
MESH_EMITTER = ('AttributeBegin\nAreaLightSource "area" "color L" [3 6 12]\nMaterial "matte" "color Kd" [0 0 0]\n'
                'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [20 150 100  20 150 300  20 350 300  20 350 100]\nAttributeEnd\n')
POINT2 = 'LightSource "point" "point from" [100 300 100] "color I" [150000 250000 150000]\n'

def _mesh(mat, xf, **kw):
    return "AttributeBegin\n%s\n%s\n%sAttributeEnd\n" % (mat, xf, scenes.smooth_mesh_text(**kw))

# per-vertex N / uv / S (trianglemesh.cpp:71-133): smooth-shaded blobs under several materials and transforms
MESH_N = (_mesh('Material "matte" "color Kd" [.7 .6 .3]', "Translate 160 110 330", radius=100, with_n=True, with_uv=False) +
          _mesh('Material "plastic" "color Kd" [.2 .3 .7] "float roughness" [.1]', "Translate 400 130 220\nRotate 40 1 0.3 0\nScale 1 1.4 0.8", radius=80, with_n=True, with_uv=False, squash=(1, 1, .7)))
MESH_NUV = (_mesh('Material "matte" "color Kd" [.6 .6 .6] "float sigma" [30]', "Translate 170 120 300\nRotate 25 0 1 0", radius=100, with_n=True, with_uv=True, squash=(1, .8, 1)) +
            _mesh('Material "plastic" "color Kd" [.6 .2 .2] "float roughness" [.2]', "ReverseOrientation\nTranslate 400 300 300\nScale -1 1 1", radius=70, with_n=True, with_uv=True, mirror_uv=True))
MESH_NSUV = (_mesh('Material "matte" "color Kd" [.5 .6 .7]', "Translate 160 110 330\nRotate 70 1 0 0", radius=95, with_n=True, with_uv=True, with_s=True) +
             _mesh('Material "glass" "float index" [1.5]', "Translate 390 150 230\nScale 1 1.2 1", radius=85, with_n=True, with_uv=True, with_s=True, squash=(1, 1, .85)))
MESH_UV = _mesh('Material "matte" "color Kd" [.7 .7 .4]', "Translate 200 130 300\nRotate 30 0 0 1", radius=110, with_n=False, with_uv=True, mirror_uv=True)
MESH_S = _mesh('Material "plastic" "color Kd" [.3 .6 .3] "float roughness" [.3]', "Translate 300 140 300\nRotate 50 1 1 0", radius=110, with_n=False, with_uv=False, with_s=True)
MESH_GM = (_mesh('Material "glass" "float index" [1.4]', "Translate 170 130 280", radius=100, with_n=True, with_uv=False, squash=(1, 1, .8)) +
           _mesh('Material "mirror"', "Translate 410 140 330\nRotate 20 0 1 0", radius=90, with_n=True, with_uv=True))

CONFIGS = {
    # name: cornell_scene kwargs  (all keyed RNG + counted rays)
    "mesh_n_whitted": dict(xres=40, yres=40, integrator="whitted", world_kwargs=dict(extra=MESH_N, point_light=True)),
    "mesh_nuv_direct": dict(xres=40, yres=40, integrator="directlighting", xsamples=2, ysamples=1, jitter=True, world_kwargs=dict(extra=MESH_NUV, light_nsamples=2)),
    "mesh_nsuv_path": dict(xres=40, yres=40, integrator="path", xsamples=2, ysamples=2, jitter=True, world_kwargs=dict(extra=MESH_NSUV)),
    "mesh_uv_only_path": dict(xres=36, yres=36, integrator="path", xsamples=2, ysamples=2, world_kwargs=dict(extra=MESH_UV)),
    "mesh_s_only_direct": dict(xres=36, yres=36, integrator="directlighting", world_kwargs=dict(extra=MESH_S)),
    "mesh_n_glass_mirror_whitted": dict(xres=40, yres=40, integrator="whitted", xsamples=2, ysamples=1, world_kwargs=dict(extra=MESH_GM)),
    "whitted_point": dict(xres=48, yres=48, integrator="whitted", world_kwargs=dict(point_light=True, area_light=False)),
    "whitted_area": dict(xres=48, yres=48, integrator="whitted"),
    "whitted_glass_mirror": dict(xres=48, yres=48, integrator="whitted", world_kwargs=dict(mirror_quad=True, glass_sphere_tris=blob)),
    "direct_all": dict(xres=48, yres=48, integrator="directlighting"),
    "direct_all_ns4_spp4": dict(xres=32, yres=32, integrator="directlighting", xsamples=2, ysamples=2, jitter=True,
                                world_kwargs=dict(light_nsamples=4)),
    "direct_one_point_and_area": dict(xres=48, yres=48, integrator="directlighting", integrator_params='"string strategy" ["one"]',
                                      world_kwargs=dict(point_light=True)),
    "direct_glass": dict(xres=40, yres=40, integrator="directlighting", world_kwargs=dict(glass_sphere_tris=blob)),
    "path_box_4spp": dict(xres=48, yres=48, integrator="path", xsamples=2, ysamples=2),
    "path_jitter_mitchell_4spp": dict(xres=40, yres=40, integrator="path", xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell"),
    "path_gaussian_depth8": dict(xres=32, yres=32, integrator="path", maxdepth=8, xsamples=2, ysamples=1, jitter=True, pixel_filter="gaussian"),
    "path_soup2k": dict(xres=48, yres=48, integrator="path", xsamples=2, ysamples=2, soup_tris=2000),
    "path_glass_mirror": dict(xres=40, yres=40, integrator="path", xsamples=2, ysamples=2, world_kwargs=dict(mirror_quad=True, glass_sphere_tris=blob)),
    "path_lens_crop": dict(xres=64, yres=64, integrator="path", xsamples=2, ysamples=2, jitter=True, lensradius=6.0, focaldistance=900.0,
                           crop=(0.25, 0.75, 0.3, 0.8)),
    "direct_soup5k_seed7": dict(xres=48, yres=48, integrator="directlighting", soup_tris=5000, seed=7, xsamples=2, ysamples=2, jitter=True),
    "ld_path_mitchell": dict(xres=32, yres=32, integrator="path", sampler="lowdiscrepancy", pixelsamples=4, pixel_filter="mitchell"),
    "ld_direct_ns3_soup": dict(xres=32, yres=32, integrator="directlighting", sampler="lowdiscrepancy", pixelsamples=3, soup_tris=800,
                               world_kwargs=dict(light_nsamples=3)),
    "ld_whitted_lens": dict(xres=32, yres=32, integrator="whitted", sampler="lowdiscrepancy", pixelsamples=8, lensradius=5.0, focaldistance=800.0),
    "random_path": dict(xres=32, yres=32, integrator="path", sampler="random", xsamples=2, ysamples=2),
    "random_whitted_lens": dict(xres=32, yres=32, integrator="whitted", sampler="random", xsamples=2, ysamples=1, lensradius=5.0, focaldistance=800.0),
    "grid_whitted_lazy": dict(xres=48, yres=48, integrator="whitted", accelerator="grid"),
    "grid_path_soup3k_eager": dict(xres=48, yres=48, integrator="path", xsamples=2, ysamples=2, soup_tris=3000, accelerator="grid",
                                   accel_params='"bool refineimmediately" ["true"]'),
    "grid_direct_glass_lazy": dict(xres=40, yres=40, integrator="directlighting", accelerator="grid",
                                   world_kwargs=dict(mirror_quad=True, glass_sphere_tris=blob)),
    "vol_emission_path": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=1, volume_integrator='"emission" "float stepsize" [40]',
                              world_kwargs=dict(volume='"color Le" [.002 .003 .004]')),
    "vol_single_whitted": dict(xres=32, yres=32, integrator="whitted", volume_integrator='"single" "float stepsize" [60]', world_kwargs=dict(volume=' ')),
    "vol_single_direct_glass": dict(xres=32, yres=32, integrator="directlighting", volume_integrator='"single" "float stepsize" [80]',
                                    world_kwargs=dict(volume='"float g" [.3]', glass_sphere_tris=blob, point_light=True)),
    "vol_single_path_grid": dict(xres=24, yres=24, integrator="path", xsamples=2, ysamples=2, jitter=True, accelerator="grid",
                                 volume_integrator='"single" "float stepsize" [50]', world_kwargs=dict(volume='"color Le" [.001 .001 .001]')),
    # spot and distant lights (also under a non-identity CTM)
    "whitted_spot_distant": dict(xres=40, yres=40, integrator="whitted", world_kwargs=dict(area_light=False, extra=SPOT + DISTANT)),
    "direct_spot_area": dict(xres=40, yres=40, integrator="directlighting", xsamples=2, ysamples=1, jitter=True, world_kwargs=dict(extra=SPOT_XF)),
    "path_distant_spot_glass": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=2, world_kwargs=dict(area_light=False, extra=SPOT + DISTANT_XF,
                                                                                                              glass_sphere_tris=blob)),
    "vol_single_spot": dict(xres=24, yres=24, integrator="whitted", volume_integrator='"single" "float stepsize" [60]',
                            world_kwargs=dict(volume=' ', area_light=False, extra=SPOT)),
    # spheres (full, clipped, transformed, reversed), matte / glass / mirror / plastic
    "sphere_whitted": dict(xres=48, yres=48, integrator="whitted", world_kwargs=dict(point_light=True, extra=SPHERES)),
    "sphere_direct": dict(xres=40, yres=40, integrator="directlighting", xsamples=2, ysamples=1, jitter=True, world_kwargs=dict(extra=SPHERES)),
    "sphere_path_grid": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=2, accelerator="grid", world_kwargs=dict(extra=SPHERES)),
    "sphere_path_soup": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=2, jitter=True, soup_tris=500, world_kwargs=dict(extra=SPHERES)),
    # disks and cylinders (full, annular, partial phi, transformed, reversed)
    "quadrics_whitted": dict(xres=48, yres=48, integrator="whitted", world_kwargs=dict(point_light=True, extra=QUADRICS)),
    "quadrics_direct_grid": dict(xres=40, yres=40, integrator="directlighting", xsamples=2, ysamples=1, jitter=True, accelerator="grid", world_kwargs=dict(extra=QUADRICS)),
    "quadrics_path": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=2, world_kwargs=dict(extra=QUADRICS)),
    # quadric emitters (sphere / disk / cylinder area lights), with and without the Cornell ceiling light
    "qlight_whitted": dict(xres=40, yres=40, integrator="whitted", world_kwargs=dict(area_light=False, extra=QLIGHTS)),
    "qlight_direct_ns2": dict(xres=32, yres=32, integrator="directlighting", xsamples=2, ysamples=1, jitter=True, world_kwargs=dict(extra=QLIGHTS.replace('"color L"', '"integer nsamples" [2] "color L"'))),
    "qlight_path": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=2, world_kwargs=dict(area_light=False, extra=QLIGHTS, mirror_quad=True)),
    # cones, paraboloids, hyperboloids
    "quadrics2_whitted": dict(xres=48, yres=48, integrator="whitted", world_kwargs=dict(point_light=True, extra=QUADRICS2)),
    "quadrics2_direct": dict(xres=40, yres=40, integrator="directlighting", xsamples=2, ysamples=1, jitter=True, world_kwargs=dict(extra=QUADRICS2)),
    "quadrics2_path_grid": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=2, accelerator="grid", world_kwargs=dict(extra=QUADRICS2)),
    # plastic (Lambertian + Blinn microfacet lobes): text substitution of two Material lines below
    "plastic_whitted": dict(xres=40, yres=40, integrator="whitted", world_kwargs=dict(point_light=True)),
    "plastic_direct_ns2": dict(xres=32, yres=32, integrator="directlighting", xsamples=2, ysamples=1, jitter=True, world_kwargs=dict(light_nsamples=2)),
    "plastic_path": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=2, jitter=True, world_kwargs=dict(glass_sphere_tris=blob)),
    # uber (T, D, G, R lobes): text substitution of two Material lines below
    "uber_whitted": dict(xres=40, yres=40, integrator="whitted", world_kwargs=dict(point_light=True)),
    "uber_direct": dict(xres=32, yres=32, integrator="directlighting", xsamples=2, ysamples=1, jitter=True),
    "uber_path": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=2, jitter=True, maxdepth=6),
    # orthographic / environment cameras: text substitution of the Camera line below
    "ortho_whitted_lens": dict(xres=40, yres=32, integrator="whitted", xsamples=2, ysamples=1, jitter=True, lensradius=6.0, focaldistance=900.0),
    "ortho_path": dict(xres=32, yres=32, integrator="path", xsamples=2, ysamples=2),
    "env_whitted": dict(xres=48, yres=24, integrator="whitted", world_kwargs=dict(point_light=True)),
    "env_path": dict(xres=32, yres=16, integrator="path", xsamples=2, ysamples=2, jitter=True),
    # DirectLighting "weighted" (WeightedSampleOneLight transport.cpp:71-122): the frame-long recurrence over every shading point in program order
    "weighted_delta3_glass": dict(xres=40, yres=40, integrator="directlighting", integrator_params='"string strategy" ["weighted"]', xsamples=2, ysamples=1, jitter=True,
                                  world_kwargs=dict(point_light=True, area_light=False, extra=SPOT + DISTANT, mirror_quad=True, glass_sphere_tris=blob)),
    "weighted_qlights_point": dict(xres=32, yres=32, integrator="directlighting", integrator_params='"string strategy" ["weighted"]', xsamples=2, ysamples=2, jitter=True,
                                   world_kwargs=dict(point_light=True, area_light=False, extra=QLIGHTS, mirror_quad=True)),
    "weighted_mesh_emitters": dict(xres=40, yres=40, integrator="directlighting", integrator_params='"string strategy" ["weighted"]', xsamples=2, ysamples=1, jitter=True,
                                   world_kwargs=dict(extra=MESH_EMITTER, glass_sphere_tris=blob)),
    # lights of MIXED RNG use (VERDICT r04 missing #3): the Cornell ceiling emitter (two triangles: ShapeSet::Sample draws one RandomFloat per estimate)
    # next to a point light and a spot (none), mirror + glass recursion: where in the stream the emitter's draw sits depends on the lights chosen before
    "weighted_cornell_plus_point": dict(xres=40, yres=40, integrator="directlighting", integrator_params='"string strategy" ["weighted"]', xsamples=2, ysamples=1, jitter=True,
                                        maxdepth=4, world_kwargs=dict(point_light=True, extra=SPOT, mirror_quad=True, glass_sphere_tris=blob)),
    "weighted_one_light": dict(xres=32, yres=32, integrator="directlighting", integrator_params='"string strategy" ["weighted"]'),
    "weighted_ld_six_lights_soup": dict(xres=32, yres=32, integrator="directlighting", integrator_params='"string strategy" ["weighted"]', maxdepth=3, sampler="lowdiscrepancy",
                                        pixelsamples=4, soup_tris=800, soup_materials=True,
                                        world_kwargs=dict(point_light=True, area_light=False, extra=SPOT + DISTANT + SPOT_XF + DISTANT_XF + POINT2)),
    "whitted_orennayar_triangle": dict(xres=32, yres=32, integrator="whitted", pixel_filter="triangle",
                                       world_kwargs=dict(point_light=True)),
}

EDGE_HDR = scenes.options_block(xres=16, yres=16, integrator="path", xsamples=2, ysamples=1, keyed=True, count=True)
EDGE_SCENES = {   # degenerate inputs: empty world, a single triangle, zero-area / repeated-index triangles, nsamples 0
    "edge_empty_world": EDGE_HDR + 'WorldBegin\nLightSource "point" "point from" [278 273 0]\nWorldEnd\n',
    "edge_one_triangle": EDGE_HDR + 'WorldBegin\nLightSource "point" "point from" [278 273 -100]\nShape "trianglemesh" "integer indices" [0 1 2] '
                         '"point P" [100 100 300 450 100 300 278 450 300]\nWorldEnd\n',
    "edge_degenerate_triangles": EDGE_HDR + 'WorldBegin\nLightSource "point" "point from" [278 273 -100]\nShape "trianglemesh" "integer indices" [0 1 2 0 0 0 3 4 5] '
                                 '"point P" [100 100 300 450 100 300 278 450 300  1 1 1 2 2 2 3 3 3]\nWorldEnd\n',
    "edge_emitter_nsamples0": EDGE_HDR.replace('"path"', '"directlighting"') + 'WorldBegin\nAreaLightSource "area" "color L" [5 5 5] "integer nsamples" [0]\n'
                              'Shape "trianglemesh" "integer indices" [0 2 1] "point P" [100 100 300 450 100 300 278 450 300]\n'
                              'AreaLightSource "area" "color L" [0 0 0]\nMaterial "matte"\nShape "trianglemesh" "integer indices" [0 1 2 0 2 3] '
                              '"point P" [0 0 500 556 0 500 556 549 500 0 549 500]\nWorldEnd\n',
}


def main():
    only = set(sys.argv[1:])
    if only:
        for k in list(EDGE_SCENES):
            if k not in only: del EDGE_SCENES[k]
        for k in list(CONFIGS):
            if k not in only: del CONFIGS[k]
    for name, text in EDGE_SCENES.items():
        rgb, alpha, st = REF.run_reference(text, keyed=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), scene=np.array(text), rgb=rgb, alpha=alpha, stats=np.array(json.dumps(st)))
        print(name, rgb.shape, "mean", float(rgb.mean()), {k: st[k] for k in ("closest_rays", "any_rays")})
    for name, kw in CONFIGS.items():
        text = scenes.cornell_scene(keyed=True, count=True, **kw)
        if name.startswith("plastic_"):
            text = text.replace('Material "matte" "color Kd" [0.73 0.73 0.73]', 'Material "plastic" "color Kd" [0.5 0.5 0.55] "color Ks" [0.4 0.4 0.4] "float roughness" [0.08]')
            text = text.replace('Material "matte" "color Kd" [0.14 0.45 0.091]', 'Material "plastic" "color Kd" [0.1 0.4 0.1] "color Ks" [0.6 0.5 0.5] "float roughness" [0.3]')
            assert text.count('"plastic"') >= 2, text[:2000]
        if name.startswith("uber_"):
            text = text.replace('Material "matte" "color Kd" [0.73 0.73 0.73]', 'Material "uber" "color Kd" [0.5 0.5 0.55] "color Ks" [0.3 0.3 0.3] "color Kr" [0.2 0.25 0.2] "float roughness" [0.12]')
            text = text.replace('Material "matte" "color Kd" [0.14 0.45 0.091]', 'Material "uber" "color Kd" [0.1 0.4 0.1] "color Ks" [0 0 0] "color opacity" [0.6 0.7 0.6]')
            text = text.replace('Material "matte" "color Kd" [0.63 0.065 0.05]', 'Material "uber" "color Kd" [0 0 0] "color Ks" [0.5 0.4 0.4] "color Kr" [0.4 0.1 0.1] "color opacity" [1 1 0.5] "float roughness" [0.3]')
            assert text.count('"uber"') >= 3, text[:3000]
        if name.startswith("ortho_"):
            text = text.replace('Camera "perspective" "float fov" [39.3]', 'Camera "orthographic" "float screenwindow" [-300 300 -290 290]')
            assert "orthographic" in text
        if name.startswith("env_"):
            text = text.replace("LookAt 278 273 -800  278 273 0  0 1 0", "LookAt 278 273 200  278 273 600  0 1 0").replace(
                'Camera "perspective" "float fov" [39.3]', 'Camera "environment"')
            assert "environment" in text
        if name == "whitted_orennayar_triangle":
            text = text.replace('Material "matte" "color Kd" [0.73 0.73 0.73]', 'Material "matte" "color Kd" [0.73 0.73 0.73] "float sigma" [35]')
        rgb, alpha, st = REF.run_reference(text, keyed=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), scene=np.array(text), rgb=rgb, alpha=alpha, stats=np.array(json.dumps(st)))
        print(name, rgb.shape, "mean", float(rgb.mean()), {k: st[k] for k in ("closest_rays", "any_rays")}, st.get("stats", {}))
    # probe fixtures: camera rays + hits
    for name, kw in {} if only else {"probe_cornell": dict(xres=24, yres=24), "probe_soup3k_jitter": dict(xres=24, yres=24, soup_tris=3000, xsamples=2, ysamples=1, jitter=True),
                     "probe_lens": dict(xres=16, yres=16, xsamples=2, ysamples=2, jitter=True, lensradius=4.0, focaldistance=700.0)}.items():
        d = tempfile.mkdtemp()
        dump = os.path.join(d, "rays.bin")
        text = scenes.cornell_scene(keyed=True, count=True, integrator="probe",
                                    integrator_params='"string dump" ["%s"] "point target" [278 540 280]' % dump, **kw)
        rgb, alpha, st = REF.run_reference(text, keyed=True, workdir=d)
        rec = np.fromfile(dump, np.float32).reshape(-1, 20)
        # the scene stored in the fixture names no dump file (the product ignores the probe integrator's params)
        text = text.replace(dump, "probe_rays.bin")
        np.savez_compressed(os.path.join(OUT, name + ".npz"), scene=np.array(text), records=rec, stats=np.array(json.dumps(st)))
        print(name, rec.shape, "hits", int(rec[:, 8].sum()), "occluded", int(rec[:, 18].sum()))

if __name__ == "__main__":
    main()
