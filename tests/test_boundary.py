"""The drop-in boundary from the reference's side.

oracle/ref/hip_adapter.cpp is a real pbrt-v1 plugin (extern "C" CreateSurfaceIntegrator + CreateAccelerator, the reference's own
plugin ABI, core/dynload.cpp:185-205) compiled against the reference's headers.  Inside the unmodified reference binary it
flattens the reference's OWN objects into RtSceneDesc / RtRenderDesc.  These tests assert that
  * the product's host front end (own parser + API state machine + factories) emits BYTE-IDENTICAL descriptors for the same scene
    text -- against committed fixtures of the adapter's output (runs anywhere) and live when oracle/_ref is present;
  * (GPU) the reference's own Scene::Render loop + ImageFilm, with Li() served by the HIP library through the C ABI, produces the
    film the product produces on its own."""
import os
import lzma
import numpy as np
import pytest
import __graft_entry__ as g_entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DESC = os.path.join(ROOT, "tests", "golden", "desc")
SMOOTH = ('AttributeBegin\nMaterial "plastic" "color Kd" [.2 .3 .7] "float roughness" [.1]\nTranslate 300 200 300\nRotate 30 1 1 0\nScale 1 1.2 .9\n%s'
          'AttributeEnd\n')


# quadrics (two of them emitters), spot + distant lights, uber / plastic materials
F4_WORLD = ('AttributeBegin\nMaterial "uber" "color Kd" [.5 .4 .6] "color Ks" [.3 .3 .3] "color Kr" [.2 .2 .2] "color opacity" [1 1 .7] "float roughness" [.2]\n'
            'Translate 420 100 380\nRotate 25 1 1 1\nShape "cylinder" "float radius" [48] "float zmin" [-50] "float zmax" [60] "float phimax" [300]\nAttributeEnd\n'
            'AttributeBegin\nAreaLightSource "area" "color L" [6 5 4] "integer nsamples" [2]\nMaterial "matte" "color Kd" [0 0 0]\nTranslate 150 420 300\n'
            'Shape "sphere" "float radius" [40] "float zmin" [-30] "float zmax" [35]\nAttributeEnd\n'
            'AttributeBegin\nAreaLightSource "area" "color L" [3 3 5]\nTranslate 400 500 150\nRotate 90 1 0 0\nReverseOrientation\nShape "disk" "float radius" [50] "float innerradius" [10]\nAttributeEnd\n'
            'AttributeBegin\nMaterial "plastic" "color Kd" [.2 .5 .3] "color Ks" [.4 .4 .4] "float roughness" [.05]\nTranslate 120 60 200\nRotate -90 1 0 0\n'
            'Shape "cone" "float radius" [50] "float height" [120]\nTranslate 180 0 0\nShape "paraboloid" "float radius" [40] "float zmax" [90] "float zmin" [10]\n'
            'Translate 0 140 0\nScale -1 1 1\nShape "hyperboloid" "point p1" [50 0 -40] "point p2" [30 30 50]\nAttributeEnd\n'
            'AttributeBegin\nRotate 10 0 1 0\nLightSource "spot" "point from" [278 500 100] "point to" [300 0 330] "color I" [300000 250000 200000] "float coneangle" [35] "float conedeltaangle" [12]\n'
            'LightSource "distant" "point from" [.3 1 -.2] "point to" [0 0 0] "color L" [.8 .8 1]\nAttributeEnd\n')


def cases(scenes):
    blob = scenes.icosphere((200, 120, 250), 90, 1)
    f4 = scenes.cornell_scene(xres=32, yres=28, integrator="path", maxdepth=4, xsamples=2, ysamples=2, jitter=True, soup_tris=200, world_kwargs=dict(extra=F4_WORLD))
    return {
        "f4_quadrics_lights_uber": f4,
        "f4_orthographic": scenes.cornell_scene(xres=24, yres=20, integrator="directlighting", xsamples=2, ysamples=1, lensradius=3.0, focaldistance=700.0,
                                                world_kwargs=dict(extra=F4_WORLD)).replace('Camera "perspective" "float fov" [39.3]', 'Camera "orthographic" "float screenwindow" [-300 300 -290 290]'),
        "f4_environment": scenes.cornell_scene(xres=32, yres=16, integrator="whitted", world_kwargs=dict(extra=F4_WORLD)).replace(
            "LookAt 278 273 -800  278 273 0  0 1 0", "LookAt 278 273 200  278 273 600  0 1 0").replace('Camera "perspective" "float fov" [39.3]', 'Camera "environment" "float hither" [.5]'),
        "c1_cornell_whitted": scenes.cornell_scene(xres=64, yres=64, integrator="whitted"),
        "c2_cornell_path": scenes.cornell_scene(xres=48, yres=48, integrator="path", maxdepth=5, xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell"),
        "c3_soup_direct": scenes.cornell_scene(xres=40, yres=30, integrator="directlighting", xsamples=2, ysamples=1, jitter=True, soup_tris=600, pixel_filter="gaussian",
                                               world_kwargs=dict(light_nsamples=3, point_light=True)),
        "c4_mix_path_ld": scenes.cornell_scene(xres=32, yres=32, integrator="path", maxdepth=8, sampler="lowdiscrepancy", pixelsamples=6, soup_tris=300, soup_materials=True,
                                               world_kwargs=dict(mirror_quad=True, glass_sphere_tris=blob), crop=(0.1, 0.9, 0.2, 0.8), lensradius=4.0, focaldistance=800.0),
        "c5_volume_single": scenes.cornell_scene(xres=24, yres=24, integrator="directlighting", integrator_params='"string strategy" ["one"]', sampler="random", xsamples=2, ysamples=2,
                                                 soup_tris=100, volume_integrator='"single" "float stepsize" [40]', world_kwargs=dict(volume='"float g" [.2] "color Le" [.01 .01 .02]'),
                                                 accelerator="grid"),
        "smooth_mesh_nuvs": scenes.cornell_scene(xres=24, yres=24, integrator="whitted", accel_params='"integer maxprims" [2] "float emptybonus" [.3]',
                                                 world_kwargs=dict(extra=SMOOTH % scenes.smooth_mesh_text(nu=6, nv=4, with_n=True, with_uv=True, with_s=True))),
    }


def sections(b):
    """{tag: bytes} of an rt_desc_serialize image (for a readable failure)."""
    out, at = {}, 0
    while at < len(b):
        tag = b[at:at + 8].rstrip(b"\0").decode(); n = int(np.frombuffer(b[at + 8:at + 16], np.uint64)[0])
        nxt = b.find(b"\0", at + 16)
        out[tag] = (n, at)
        # sizes are implied by the tag order; walk by searching the next known tag
        at += 16
        rest = [b.find(t.encode().ljust(8, b"\0"), at) for t in ("TRIMAT", "TRILIGHT", "TRIFLAGS", "MATERIAL", "LIGHTS", "LIGHTTRI", "CAMERA", "VOLUME", "ACCEL", "QUADRICS",
                                                                "TRISHIDX", "SHADING", "XFORMS", "RENDER")]
        rest = [r for r in rest if r >= at]
        if tag == "RENDER" or not rest: break
        at = min(rest)
    return out


def differing_sections(a, b):
    sa, sb = sections(a), sections(b)
    tags = list(sa)
    bad = []
    for i, t in enumerate(tags):
        ea = sa[tags[i + 1]][1] if i + 1 < len(tags) else len(a)
        eb = sb.get(tags[i + 1], (0, len(b)))[1] if i + 1 < len(tags) else len(b)
        if t not in sb or a[sa[t][1]:ea] != b[sb[t][1]:eb]: bad.append(t)
    return bad


def test_host_front_end_emits_the_adapters_descriptors(pkg, scenes):
    """Committed fixtures: tests/golden/desc/<case>.xz = what hip_adapter.cpp produced inside the reference (make_desc_fixtures)."""
    for name, text in cases(scenes).items():
        ps = pkg.ParsedScene(text=text)
        assert ps.valid and ps.errors == 0, name
        mine = ps.serialize()
        ref = lzma.decompress(open(os.path.join(DESC, name + ".xz"), "rb").read())
        assert mine == ref, (name, differing_sections(mine, ref))


def test_adapter_live_when_the_reference_is_built(pkg, scenes):
    rr = g_entry.load_ref_runner()
    if not os.path.exists(os.path.join(rr.REF_DIR, "bin", "hip.so")):
        pytest.skip("oracle/_ref/bin/hip.so not on this box")
    for name, text in cases(scenes).items():
        ref = rr.reference_descriptors(text)
        mine = pkg.ParsedScene(text=text).serialize()
        assert mine == ref, (name, differing_sections(mine, ref))


def test_host_library_exports_a_c_surface_only(pkg):
    """libpbrt_host.so's dynamic symbol table: the pbrt_host_* C entry points and nothing of the C++ implementation."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.HOST_LIB], capture_output=True, text=True).stdout
    syms = [ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TDBRW"]
    assert syms and all(s.startswith("pbrt_host_") for s in syms), [s for s in syms if not s.startswith("pbrt_host_")][:10]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c1_cornell_whitted", "c3_soup_direct", "c2_cornell_path", "smooth_mesh_nuvs", "f4_orthographic", "f4_environment"])
def test_reference_render_loop_served_by_the_hip_library(pkg, scenes, name):
    """Scene::Render (scene.cpp:32-88) of the UNMODIFIED reference, with SurfaceIntegrator "hip" / Accelerator "hip": the adapter hands
    the flattened scene to libpbrt_hip.so through the C ABI and answers every Li() from rt_samples_read; the reference's sampler
    (keyed), ImageFilm::AddSample and WriteImage do the rest.  The film must equal the one the product renders on its own:
    bit for bit for Whitted / DirectLighting (identical samples, identical film arithmetic), to 1e-5 for the path integrator."""
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible")
    rr = g_entry.load_ref_runner()
    if not os.path.exists(os.path.join(rr.REF_DIR, "bin", "hip.so")):
        pytest.skip("oracle/_ref/bin/hip.so not on this box")
    import re
    text = cases(scenes)[name]
    text = re.sub(r'Sampler "(\w+)"', r'Sampler "keyed" "string inner" ["\1"] "integer seed" [3]', text)
    rgb_ref, alpha_ref, _ = rr.run_reference(rr.as_hip_plugin_scene(text), keyed=True, env={"PBRT_HIP_LIB": pkg.HIP_LIB})
    rgb, alpha, cnt, _ = pkg.render_text(text)
    assert cnt["camera_rays"] > 0
    tol = 1e-5 if "path" in name else 0.0
    assert np.abs(rgb - rgb_ref).max() <= tol and np.abs(alpha - alpha_ref).max() <= tol, (name, float(np.abs(rgb - rgb_ref).max()))


def make_desc_fixtures():
    """Authoring container only: python -c 'import tests.test_boundary as t; t.make_desc_fixtures()' (needs oracle/_ref)."""
    pkg = g_entry.load_package()
    from pbrt_v1_amd import scenes
    rr = g_entry.load_ref_runner()
    os.makedirs(DESC, exist_ok=True)
    for name, text in cases(scenes).items():
        b = rr.reference_descriptors(text)
        open(os.path.join(DESC, name + ".xz"), "wb").write(lzma.compress(b, preset=9))
        print(name, len(b), "bytes")
