"""Host front end (pbrt-v1_amd/csrc/host): scene parser, ParamSet semantics, API state machine, plugin factory
defaults.  Reference behaviours cited per test.  CPU only."""
import os
import numpy as np
import pytest

BASE = '''LookAt 0 0 -5  0 0 0  0 1 0
Camera "perspective" "float fov" [45]
Film "image" "integer xresolution" [16] "integer yresolution" [8]
Sampler "stratified"
%s
WorldBegin
%s
WorldEnd
'''
TRI = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1 -1 0  1 -1 0  0 1 0]\n'


def test_defaults_follow_the_reference_factories(pkg):
    ps = pkg.ParsedScene(text=BASE % ("", 'LightSource "point"\n' + TRI))
    assert ps.valid and ps.errors == 0
    # api.cpp:62-71 defaults: mitchell 2x2 (mitchell.cpp:54-57) => extent = res + 4; stratified 2x2 => 4 spp;
    # directlighting strategy all (directlighting.cpp:198)
    assert ps.sample_extent == (-2, 18, -2, 10) and ps.spp == 4 and ps.integrator == 1
    assert (ps.width, ps.height) == (16, 8) and ps.premultiply
    assert ps.n_tris == 1 and ps.n_lights == 1 and ps.n_materials == 1


def test_box_filter_extent_and_crop(pkg):
    ps = pkg.ParsedScene(text=(BASE % ('PixelFilter "box"\nSampler "stratified" "integer xsamples" [3] "integer ysamples" [1]', TRI))
                         .replace('[8]', '[8] "float cropwindow" [.25 .75 0 .5]'))
    # image.cpp:79-86 crop in pixels, :148-156 sample extent = floor(start + .5 -/+ width)
    assert (ps.width, ps.height) == (8, 4) and ps.sample_extent == (4, 13, 0, 5) and ps.spp == 3
    assert ps.warnings >= 1          # no lights: scene.cpp:112-114 warning


def test_camera_matrices(pkg):
    ps = pkg.ParsedScene(text=BASE % ("", TRI))
    r2c, c2w = ps.camera_matrices()
    # LookAt from (0,0,-5) towards +z: camera-to-world translation column (transform.cpp:113-138)
    assert np.allclose(c2w[:3, 3], [0, 0, -5]) and np.allclose(c2w[3], [0, 0, 0, 1])
    assert np.allclose(c2w[:3, :3] @ c2w[:3, :3].T, np.eye(3), atol=1e-6)
    # raster (8,4) is the image centre: it maps onto the optical axis (x = y = 0 after the w divide)
    p = r2c @ np.array([8, 4, 0, 1], np.float32)
    assert abs(p[0] / p[3]) < 1e-6 and abs(p[1] / p[3]) < 1e-6


def test_paramset_semantics(pkg):
    # ints are floats truncated (pbrtparse.y:484-492); a bare string for a colour re-types it as a texture
    # (pbrtparse.y:476-480) -> unknown texture name -> Error + default (paramset.cpp:434-449)
    world = 'Material "matte" "color Kd" "nosuchtex"\n' + TRI.replace("[0 1 2]", "[0.9 1.2 2.7]")
    ps = pkg.ParsedScene(text=BASE % ("", world))
    assert ps.n_tris == 1 and ps.errors >= 1
    # misspelt parameter -> "not used" warning (paramset.cpp:242-254), still renders
    ps2 = pkg.ParsedScene(text=BASE % ('SurfaceIntegrator "path" "integer maxdepht" [3]', 'LightSource "point"\n' + TRI))
    assert ps2.valid and ps2.warnings >= 1 and ps2.integrator == 2


def test_state_machine_errors(pkg):
    # options inside the world block are rejected and ignored (api.cpp:120-139)
    ps = pkg.ParsedScene(text=BASE % ("", 'Sampler "stratified" "integer xsamples" [7]\nLightSource "point"\n' + TRI))
    assert ps.errors >= 1 and ps.spp == 4
    # unmatched AttributeEnd (api.cpp:282-287)
    ps = pkg.ParsedScene(text=BASE % ("", "AttributeEnd\nLightSource \"point\"\n" + TRI))
    assert ps.errors >= 1 and ps.n_tris == 1


def test_attribute_stack_and_transforms(pkg):
    world = ('AttributeBegin\nTranslate 10 0 0\nReverseOrientation\n' + TRI + 'AttributeEnd\n' + TRI + 'LightSource "point"\n')
    ps = pkg.ParsedScene(text=BASE % ("", world))
    v = ps.tri_verts()
    assert np.allclose(v[0, :, 0] - v[1, :, 0], 10) and ps.n_tris == 2


def test_primitive_order_is_kdtree_refinement_order(pkg):
    # FullyRefine pops the last refined triangle first (primitive.cpp:40-53): a 2-triangle mesh comes out reversed
    world = 'LightSource "point"\nShape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [0 0 0  1 0 0  1 1 0  0 1 0]\n'
    v = pkg.ParsedScene(text=BASE % ("", world)).tri_verts()
    assert np.allclose(v[0], [[0, 0, 0], [1, 1, 0], [0, 1, 0]]) and np.allclose(v[1], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])


def test_unsupported_plugins_are_errors_not_silent(pkg):
    ps = pkg.ParsedScene(text=BASE % ("", 'LightSource "point"\nShape "heightfield" "integer nu" [2]\n' + TRI))
    assert ps.errors >= 1 and ps.n_tris == 1
    ps = pkg.ParsedScene(text=BASE % ('Camera "fisheye"', 'LightSource "point"\n' + TRI))
    assert not ps.valid
    ps = pkg.ParsedScene(text=BASE % ("", 'Material "substrate"\nLightSource "point"\n' + TRI))
    assert ps.errors >= 1 and ps.n_materials == 1        # falls back to matte (api.cpp:376-379)
    ps = pkg.ParsedScene(text=BASE % ("", 'LightSource "goniometric"\n' + TRI))
    assert ps.errors >= 1 and ps.n_lights == 0


def test_f4_plugins_are_accepted(pkg):
    """sphere, spot / distant, plastic, orthographic / environment parse without errors."""
    world = ('LightSource "spot" "point from" [0 5 0] "point to" [0 0 0]\nLightSource "distant"\nMaterial "plastic" "float roughness" [.2]\n'
             'AttributeBegin\nTranslate 1 2 3\nShape "sphere" "float radius" [2] "float zmax" [1]\nAttributeEnd\n'
             'Shape "disk" "float radius" [.5] "float height" [1]\nShape "cylinder" "float radius" [.25]\n' + TRI)
    for cam in ('Camera "orthographic"', 'Camera "environment"', ""):
        ps = pkg.ParsedScene(text=BASE % (cam, world))
        assert ps.valid and ps.errors == 0
        assert ps.n_tris == 4 and ps.n_lights == 2            # every quadric is one primitive slot
        v = ps.tri_verts()[0]                                 # its slot = world bound {pMin, pMax, pMin}
        assert np.allclose(v[0], [-1, 0, 1]) and np.allclose(v[1], [3, 4, 4]) and np.array_equal(v[0], v[2])
    # a quadric under AreaLightSource becomes an area light whose shape is the quadric itself (area.cpp:38-39)
    ps = pkg.ParsedScene(text=BASE % ("", 'AreaLightSource "area" "color L" [3 3 3]\nShape "sphere" "float radius" [1]\n' + TRI))
    assert ps.valid and ps.errors == 0 and ps.n_lights == 2 and ps.n_light_tris == 1


def test_reference_features_off_the_path_invalidate_the_frame(pkg):
    """Nothing is silently replaced by something else: the reference's default sampler
    (bestcandidate, api.cpp:62-71) is an Error that leaves the frame invalid, exactly like an unknown plugin.  The "weighted" light
    strategy (transport.cpp:71-122) parses since round 4: rt_render runs its recurrence on the device (tests/test_gpu_weighted.py)."""
    ps = pkg.ParsedScene(text=(BASE % ("", 'LightSource "point"\n' + TRI)).replace('Sampler "stratified"\n', ""))
    assert not ps.valid and ps.errors >= 1
    ps = pkg.ParsedScene(text=BASE % ('SurfaceIntegrator "directlighting" "string strategy" ["weighted"]', 'LightSource "point"\n' + TRI))
    assert ps.valid and ps.errors == 0 and ps.warnings == 0 and ps.render_view()["strategy"] == 2
    ps = pkg.ParsedScene(text=BASE % ('SurfaceIntegrator "directlighting" "string strategy" ["wieghted"]', 'LightSource "point"\n' + TRI))
    assert ps.valid and ps.warnings == 1 and ps.render_view()["strategy"] == 0          # unknown strategy: Warning, "all" (directlighting.cpp:203-206)
    ps = pkg.ParsedScene(text=BASE % ('SurfaceIntegrator "directlighting" "string strategy" ["one"]', 'LightSource "point"\n' + TRI))
    assert ps.valid and ps.errors == 0 and ps.warnings == 0


def test_trianglemesh_vertex_data_and_the_factory_checks(pkg):
    """"uv" / "st", "N", "S" reach the scene description as per-triangle records (RtTriShading); the checks of
    shapes/trianglemesh.cpp:357-393 are applied: short uv arrays and N / S of the wrong length are discarded with an Error,
    degenerate uvs next to normals discard all uvs with a Warning."""
    import ctypes as C
    quad = ('Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [0 0 0  1 0 0  1 1 0  0 1 0] %s\n')
    def records(extra):
        ps = pkg.ParsedScene(text=BASE % ("", 'LightSource "point"\nAttributeBegin\nTranslate 0 0 1\n' + quad % extra + 'AttributeEnd\n' + TRI))
        return ps, pkg.shading_records(ps)
    ps, rec = records('"normal N" [0 0 1  0 0 1  0 1 1  1 0 1] "float uv" [0 0  1 0  1 1  0 1]')
    assert ps.valid and ps.errors == 0 and ps.warnings == 0 and ps.n_tris == 3
    idx, sh, xf = rec
    assert list(idx) == [0, 1, -1] and len(sh) == 2 and xf.shape == (1, 32)
    assert sh["flags"][0] == 3 and np.allclose(sh["uv"][0], [0, 0, 1, 1, 0, 1]) and np.allclose(sh["n"][1][6:9], [0, 1, 1])   # last triangle first
    assert np.allclose(xf[0, :16].reshape(4, 4)[:3, 3], [0, 0, 1])
    ps, rec = records('"float st" [0 0 1 0] "vector S" [1 0 0  1 0 0  1 0 0  1 0 0]')
    assert ps.errors >= 1 and rec[1]["flags"][0] == 4 and np.allclose(rec[1]["uv"][0], [0, 0, 1, 0, 1, 1])             # uvs discarded -> GetUVs defaults
    ps, rec = records('"normal N" [0 0 1 0 0 1]')
    assert ps.errors >= 1 and rec[0] is None
    ps, rec = records('"normal N" [0 0 1  0 0 1  0 0 1  0 0 1] "float uv" [0 0  0 0  1 1  0 1]')
    assert ps.errors == 0 and ps.warnings >= 1 and rec[1]["flags"][0] == 2


def test_include_and_comments(pkg, tmp_path):
    (tmp_path / "geom.pbrt").write_text("# a comment\n" + TRI)
    (tmp_path / "main.pbrt").write_text(BASE % ("", 'LightSource "point" # trailing\nInclude "geom.pbrt"\n'))
    ps = pkg.ParsedScene(path=str(tmp_path / "main.pbrt"))
    assert ps.valid and ps.n_tris == 1


def test_soup_generator_is_deterministic_lcg(scenes):
    a = scenes.lcg_soup(7)
    s = 12345
    vals = []
    for _ in range(12):
        s = (s * 1664525 + 1013904223) % (1 << 32)
        vals.append(np.float32((s >> 8) / float(1 << 24)))
    c = np.float32(50) + np.array(vals[:3], np.float32) * np.array([450, 400, 450], np.float32)
    v0 = c + (np.array(vals[3:6], np.float32) * np.float32(8) - np.float32(4))
    assert np.allclose(a[0, 0], v0) and a.shape == (7, 3, 3)
    assert a.min() >= 46 and a.max() <= 504


def test_exr_writer_round_trip_and_half_rounding(pkg, tmp_path):
    """Own minimal EXR writer (the reference delegates to OpenEXR, core/exrio.cpp:75-96): RGBA half, data window inside
    the display window; float->half must round to nearest even exactly like IEEE binary16 (numpy float16)."""
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.uniform(0, 30, 4000), rng.uniform(0, 1e-5, 500), [0, 1, 65504, 65520, 1e6, 6.1e-5, 5.96e-8, 2.98e-8, 0.333251953125]])
    vals = np.resize(vals.astype(np.float32), (50, 91, 3))
    alpha = rng.uniform(0, 1, (50, 91)).astype(np.float32)
    p = str(tmp_path / "t.exr")
    pkg.write_exr(p, vals, alpha, total_res=(200, 100), offset=(17, 23))
    rgb, a, meta = pkg.read_exr(p)
    assert meta == dict(total_res=(200, 100), offset=(17, 23))
    with np.errstate(over="ignore"):
        assert np.array_equal(rgb, vals.astype(np.float16).astype(np.float32))
        assert np.array_equal(a, alpha.astype(np.float16).astype(np.float32))
    raw = open(p, "rb").read()
    assert raw[:8] == bytes([0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0]) and b"dataWindow\x00box2i" in raw and b"displayWindow" in raw


def test_exr_reader_refuses_truncated_and_foreign_files(pkg, tmp_path):
    """every size / offset in the file is checked; nothing is read out of bounds, foreign files are refused."""
    rgb = np.random.default_rng(2).uniform(0, 1, (9, 13, 3)).astype(np.float32); alpha = np.ones((9, 13), np.float32)
    p = str(tmp_path / "a.exr"); pkg.write_exr(p, rgb, alpha, total_res=(20, 20), offset=(3, 4))
    raw = open(p, "rb").read()
    bad = {"trunc_header": raw[:40], "trunc_table": raw[:raw.index(b"screenWindowWidth") + 30], "trunc_rows": raw[:-50],
           "piz": raw.replace(b"compression\0compression\0\x01\0\0\0\0", b"compression\0compression\0\x01\0\0\0\x04"),
           "huge_window": raw.replace(np.array([3, 4, 15, 12], np.int32).tobytes(), np.array([3, 4, 2000000000, 12], np.int32).tobytes()),
           "tiled_flag": raw[:5] + b"\x02" + raw[6:], "empty": b""}
    off_table = raw.index(b"screenWindowWidth\0float\0") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1
    bad["wild_offset"] = raw[:off_table] + np.array([1 << 40], np.uint64).tobytes() + raw[off_table + 8:]
    for name, data in bad.items():
        q = str(tmp_path / (name + ".exr")); open(q, "wb").write(data)
        with pytest.raises(IOError):
            pkg.read_exr(q)
    assert pkg.read_exr(p)[2] == dict(total_res=(20, 20), offset=(3, 4))


def test_exr_reader_opens_a_file_written_by_the_openexr_library(pkg, tmp_path):
    """The one EXR file in this image that the product's writer did not produce: CPython's imghdr test datum (tests/golden/third_party),
    an RGBA-half scanline file written by the OpenEXR library (channels A B G R in the library's alphabetical order, no compression), with
    the same picture as an 8-bit PPM beside it.  The reader must decode it to the PPM's values within half precision, and the product's
    writer must reproduce the file's pixel data byte for byte (same channel order, same half encoding, same scanline layout)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "third_party")
    rgb, alpha, info = pkg.read_exr(os.path.join(here, "python.exr"))
    assert rgb.shape == (16, 16, 3) and info == dict(total_res=(16, 16), offset=(0, 0))
    head, dims, mx, data = open(os.path.join(here, "python.ppm"), "rb").read().split(b"\n", 3)
    assert head == b"P6" and dims == b"16 16" and mx == b"255"
    ppm = np.frombuffer(data, np.uint8).reshape(16, 16, 3) / 255.0
    assert np.abs(rgb - ppm).max() < 4.9e-4                                   # half has 11 significant bits: 2^-11 relative
    assert alpha.min() == 0.0 and alpha.max() == 1.0 and 0.2 < alpha.mean() < 1.0   # the logo on a transparent ground
    # write what was read: header attributes, offset table and scanline blocks come out byte for byte as the library wrote them
    q = str(tmp_path / "again.exr"); pkg.write_exr(q, rgb, alpha)
    assert open(os.path.join(here, "python.exr"), "rb").read() == open(q, "rb").read()


def test_exr_assemble_merges_crop_windows(pkg, tmp_path):
    """tools/exrassemble.cpp:42-75: crop-window images (data window inside a common display window) -> one full image."""
    rng = np.random.default_rng(3)
    full = rng.uniform(0, 4, (30, 40, 3)).astype(np.float32); fa = rng.uniform(0, 1, (30, 40)).astype(np.float32)
    paths = []
    for k, (x0, x1, y0, y1) in enumerate([(0, 20, 0, 15), (20, 40, 0, 15), (0, 40, 15, 30)]):
        p = str(tmp_path / ("tile%d.exr" % k)); paths.append(p)
        pkg.write_exr(p, full[y0:y1, x0:x1], fa[y0:y1, x0:x1], total_res=(40, 30), offset=(x0, y0))
    out = str(tmp_path / "full.exr")
    assert pkg.assemble_exr(paths, out) == 1.0
    rgb, a, meta = pkg.read_exr(out)
    assert meta == dict(total_res=(40, 30), offset=(0, 0))
    assert np.array_equal(rgb, full.astype(np.float16).astype(np.float32)) and np.array_equal(a, fa.astype(np.float16).astype(np.float32))
    assert abs(pkg.assemble_exr(paths[:2], out) - 0.5) < 1e-6 and np.all(pkg.read_exr(out)[0][15:] == 0)


def test_bench_reports_a_stale_profile_as_stale(pkg, tmp_path):
    """roofline.traffic / valu_issue come from committed rocprofv3 summaries; each carries the code_id (sha256 of the
    device code objects) of the library that was profiled, and bench.py prints the counters only when the library it loaded has the same one."""
    import importlib.util, json
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    cid = pkg.code_id()
    assert len(cid) == 16 and cid == pkg.code_id()
    prof = {"code_id": cid, "hbm_bytes_per_launch_fetch_doubled": 2e9, "hbm_bytes_per_launch_uncorrected": 1e9,
            "pmc": {"SQ_INSTS_VALU": 5.1e9}, "derived": {"VALUBusy_percent": 50.0, "VALUUtilization_percent_active_lanes": 70.0, "L2_hit_rate": 0.5}}
    json.dump(prof, open(tmp_path / "latest_w_render_kernel.json", "w"))
    out = {"roofline": {"traffic": None, "kernel_ms": 10.0}}
    bench.attach_profile(out, "w", 4e9, cid, profiles_dir=str(tmp_path))
    assert out["roofline"]["traffic"] == 2e9 and 0 < out["roofline"]["valu_issue"]["frac"] <= 1.0      # 5.1e9 inst / 1.02e12 per s = 5 ms of 10
    out = {"roofline": {"traffic": None, "kernel_ms": 10.0}}
    bench.attach_profile(out, "w", 4e9, "0123456789abcdef", profiles_dir=str(tmp_path))                # another library: nothing is printed but the reason
    assert out["roofline"]["traffic"] is None and "valu_issue" not in out["roofline"] and "stale" in out["roofline"]["traffic_source"]


def test_plugins_are_loaded_from_the_search_path(pkg, tmp_path):
    """north_star: "host-side C++ keeps pbrt's plugin API (SurfaceIntegrator/Aggregate/Sampler)".  As core/dynload.cpp:462-514 does, the host library
    resolves `SurfaceIntegrator "name"` / `Accelerator "name"` / `Sampler "name"` to name.so along the search path (SearchPath directive,
    PBRT_HIP_PLUGIN_PATH) and calls its extern "C" PbrtHipCreate<Kind> factory (include/pbrt_hip_plugin.h); the compiled-in plugins are the fall-back.
    A shared object of a plugin's name that is NOT a plugin of this library -- the reference's own stratified.so / kdtree.so on a SearchPath that points
    at a pbrt-v1 install: another ABI behind the same Create<Kind> names -- is never called: a warning at most, the built-in answers."""
    import subprocess, ctypes as C
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "plug.cpp"
    src.write_text('''
#include "pbrt_hip_plugin.h"
extern "C" int PbrtHipCreateSurfaceIntegrator(const PbrtHipParams *p, const PbrtHipParamsApi *api, PbrtHipSurfaceIntegrator *out) {
    out->kind = RT_INTEGRATOR_PATH; out->max_depth = api->find_int(p, "bounces", 3) + 1; out->strategy = RT_STRATEGY_ALL; return 0; }
extern "C" int PbrtHipCreateAccelerator(const PbrtHipParams *p, const PbrtHipParamsApi *api, PbrtHipAccelerator *out) {
    RtAccelParams a = {}; a.kind = RT_ACCEL_KDTREE; a.isect_cost = api->find_int(p, "cost", 80); a.trav_cost = 1; a.empty_bonus = 0.5f; a.max_prims = 4; a.max_depth = -1;
    out->params = a; return 0; }
extern "C" int PbrtHipCreateSampler(const PbrtHipParams *p, const PbrtHipParamsApi *api, PbrtHipSampler *out) {
    out->kind = RT_SAMPLER_STRATIFIED; out->xsamples = out->ysamples = api->find_int(p, "side", 2); out->jitter = api->find_bool(p, "jitter", 0);
    out->pixelsamples = 4; out->seed = 11; return 0; }
''')
    for name in ("bouncy", "fatleaves", "square"):
        subprocess.check_call(["g++", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / (name + ".so"))])
    from pbrt_v1_amd import scenes
    base = scenes.cornell_scene(xres=16, yres=16, integrator="path", maxdepth=5, xsamples=2, ysamples=2)
    text = ('SearchPath "%s"\n' % tmp_path) + base.replace('SurfaceIntegrator "path"', 'SurfaceIntegrator "bouncy" "integer bounces" [6] #').replace(
        'Accelerator "kdtree"', 'Accelerator "fatleaves" "integer cost" [40] #').replace('Sampler "stratified"', 'Sampler "square" "integer side" [3] "bool jitter" ["true"] #')
    assert "bouncy" in text and "fatleaves" in text and "square" in text
    ps = pkg.ParsedScene(text=text)
    assert ps.valid and ps.errors == 0 and ps.warnings == 0
    rd = ps.render_view()
    assert rd["integrator"] == 2 and rd["max_depth"] == 7 and rd["x_samples"] == 3 and rd["y_samples"] == 3 and rd["jitter"] == 1 and rd["seed"] == 11
    acc = ps.accel_params()
    assert acc["max_prims"] == 4 and acc["isect_cost"] == 40
    # a parameter the plugin never looks up is reported like any unused parameter (paramset.cpp:330-346)
    ps2 = pkg.ParsedScene(text=text.replace('"integer bounces" [6]', '"integer bounces" [6] "float nonsense" [1]'))
    assert ps2.valid and ps2.warnings == 1
    # no such file anywhere: the compiled-in plugins answer, and an unknown name is the reference's Error
    ps3 = pkg.ParsedScene(text=('SearchPath "%s"\n' % tmp_path) + base)
    assert ps3.valid and ps3.errors == 0 and ps3.render_view()["integrator"] == 2 and ps3.render_view()["max_depth"] == 5
    ps4 = pkg.ParsedScene(text=base.replace('SurfaceIntegrator "path"', 'SurfaceIntegrator "nosuchplugin"'))
    assert ps4.errors >= 1
    # objects with the reference's factory names and the reference's (C++) signatures under the built-ins' names: one that loads, one that does not
    # (an unresolved core symbol, as a real pbrt-v1 plugin has here).  Neither factory is called (they would abort); no error; the scene is the built-in's.
    ref = tmp_path / "refdir"; ref.mkdir()
    (tmp_path / "ref.cpp").write_text('''
#include <cstdlib>
struct ParamSet; struct Film;
extern "C" void *CreateSampler(const ParamSet &, const Film *) { abort(); }
extern "C" void *CreateSurfaceIntegrator(const ParamSet &) { abort(); }
extern "C" void *CreateAccelerator(const void *, const ParamSet &) { abort(); }
''')
    (tmp_path / "ref2.cpp").write_text('''
#include <cstdlib>
extern int pbrt_core_symbol_that_is_not_here;
extern "C" void *CreateVolumeIntegrator(const void *) { if (pbrt_core_symbol_that_is_not_here) abort(); return 0; }
''')
    for name in ("stratified", "path", "kdtree"):
        subprocess.check_call(["g++", "-shared", "-fPIC", str(tmp_path / "ref.cpp"), "-o", str(ref / (name + ".so"))])
    subprocess.check_call(["g++", "-shared", "-fPIC", str(tmp_path / "ref2.cpp"), "-o", str(ref / "emission.so")])
    ps5 = pkg.ParsedScene(text=('SearchPath "%s"\n' % ref) + base.replace('WorldBegin', 'VolumeIntegrator "emission"\nWorldBegin'))
    assert ps5.valid and ps5.errors == 0 and ps5.render_view()["integrator"] == 2 and ps5.render_view()["max_depth"] == 5
    assert ps5.warnings <= 1                                        # at most the one "does not load here" of emission.so
    # a plugin built against another plugin ABI is refused by its PbrtHipPluginAbi
    (tmp_path / "old.cpp").write_text('''
#include "pbrt_hip_plugin.h"
extern "C" int PbrtHipPluginAbi(void) { return PBRT_HIP_PLUGIN_ABI + 1; }
extern "C" int PbrtHipCreateSurfaceIntegrator(const PbrtHipParams *, const PbrtHipParamsApi *, PbrtHipSurfaceIntegrator *out) { out->kind = RT_INTEGRATOR_WHITTED; out->max_depth = 1; out->strategy = 0; return 0; }
''')
    old = tmp_path / "olddir"; old.mkdir()
    subprocess.check_call(["g++", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), str(tmp_path / "old.cpp"), "-o", str(old / "path.so")])
    ps6 = pkg.ParsedScene(text=('SearchPath "%s"\n' % old) + base)
    assert ps6.valid and ps6.errors == 0 and ps6.render_view()["integrator"] == 2 and ps6.warnings == 1


def test_bench_scene_texts_for_the_tuned_tree_and_the_parity_leg(pkg):
    """bench.py derives two more texts from a workload's scene: `<w>_tuned` appends KdTreeAccel's tuned build parameters to the Accelerator line (the host front
    end must take them: accelerators/kdtree.cpp:489-498's names), and the oracle-side text of the cpu_baseline / parity legs adds the crop window and wraps the sampler and
    the accelerator in the checker's helper plugins -- which scenes.for_product unwraps again, so the product parses the same frame (same crop, same tree parameters)."""
    import bench
    text, label, crop = bench.workload("tsmall")
    tuned = bench.accel_with_params(text, bench.TUNED_ACCEL)
    ps0, ps1 = pkg.ParsedScene(text=text), pkg.ParsedScene(text=tuned)
    assert ps0.valid and ps1.valid and ps0.errors == ps1.errors == 0 and ps1.warnings == ps0.warnings
    a0, a1 = ps0.accel_params(), ps1.accel_params()
    assert (a0["isect_cost"], a0["trav_cost"], a0["max_prims"]) == (80, 1, 1) and abs(a0["empty_bonus"] - 0.5) < 1e-7
    assert (a1["isect_cost"], a1["trav_cost"], a1["max_prims"]) == (2, 1, 4) and a1["empty_bonus"] == 0.0
    assert ps0.n_tris == ps1.n_tris and (ps0.width, ps0.height) == (ps1.width, ps1.height)
    keyed = bench.oracle_side_text("tsmall", crop, keyed=True)
    assert 'Sampler "keyed" "string inner" ["stratified"] "integer seed" [0]' in keyed and 'Accelerator "countaccel" "string inner" ["kdtree"]' in keyed and '"float cropwindow"' in keyed
    pk = pkg.ParsedScene(text=keyed)                       # ParsedScene unwraps the helper plugins (scenes.for_product)
    assert pk.valid and pk.errors == 0 and pk.n_tris == ps0.n_tris and pk.spp == ps0.spp
    assert pk.width == int(np.ceil(160 * crop[1])) - int(np.ceil(160 * crop[0])) and pk.height == int(np.ceil(120 * crop[3])) - int(np.ceil(120 * crop[2]))
    assert bench.workload("tsmall_tuned")[0] == tuned


def _walk_leaf(tn_x, tn_y, entries, stride, runs):
    """A leaf's record slots in test order, decoded exactly as the kernels do (rt_traverse.h leaf_cursor_enter / leaf_test_flat)."""
    MORE, LIST, POS, NONE = 0x80000000, 0x40000000, 0x3fffffff, 0xffffffff
    cur = (int(tn_x) >> 2) | (int(tn_y) & ~POS & 0xffffffff)
    cursor = (int(tn_y) & POS) << ((int(tn_y) >> 30) & 1)
    out = []
    while cur != NONE:
        pos = cur & POS
        assert pos % stride == 0
        out.append(pos // stride)
        if runs:
            cur = cur + stride if cursor > 1 else NONE
            cursor -= 1
        elif cur >= (MORE | LIST):
            cur = int(entries[cursor]); cursor += 1
        elif cur & MORE:
            cur = cursor                                     # a leaf of two: the cursor is the second entry (its last)
        else:
            cur = NONE
    return out


@pytest.mark.parametrize("n_tris,runs,copies", [(0, False, False), (3000, False, False), (3000, True, False), (3000, False, True), (40000, False, False)])
def test_leaf_entries_enumerate_the_reference_lists(pkg, scenes, n_tris, runs, copies):
    """Round 6: the flat traversal reads ONE record per primitive and walks a leaf through entries (include/pbrt_hip.h rt_accel_leaf_layout, rt_leaf_entries.h).  On the host alone:
    decoded the way the kernels decode them, every leaf of the tree must enumerate exactly the primitives the reference's leaf lists (kdtree.cpp:55-64) hold, in their order; the
    records sit where the depth-first leaf walk FIRST meets each primitive (slot k = the k-th distinct primitive of that walk); lists start at even indices; the runs form gives
    every leaf consecutive slots; the copies form one slot per reference.  40 000 triangles: > 2^20 nodes, the passes run on several threads."""
    ps = pkg.ParsedScene(text=scenes.cornell_scene(xres=4, yres=4, integrator="whitted", soup_tris=n_tris))
    tv = np.ascontiguousarray(ps.tri_verts(), np.float32).reshape(-1, 9)
    nodes, refs, tn, slot_prim, entries, stride = pkg.leaf_layout(tv, runs=runs, copies=copies)
    assert stride in (3, 4) and tn.shape == nodes.shape
    leaf = (nodes[:, 0] & 3) == 3
    assert np.array_equal(tn[~leaf], nodes[~leaf]) and np.all((tn[leaf, 0] & 3) == 3)          # the tree's shape is untouched
    idx = np.nonzero(leaf)[0]
    # the depth-first leaf walk: nodes are stored depth-first, so node order is walk order
    def ref_list(i):
        n = int(nodes[i, 0]) >> 2
        return [int(nodes[i, 1])] if n == 1 else [int(r) for r in refs[int(nodes[i, 1]):int(nodes[i, 1]) + n]]
    if not (runs or copies):
        seen, order = set(), []
        for i in idx:
            for p in ref_list(i):
                if p not in seen:
                    seen.add(p); order.append(p)
        assert order == [int(p) for p in slot_prim] and len(order) == len(tv)                  # first-touch order, every primitive exactly once
    else:
        assert len(slot_prim) == int((nodes[leaf, 0] >> 2).sum())                                # one record per reference
    rng = np.random.default_rng(1)
    sample = idx if len(idx) <= 30000 else np.concatenate([idx[:5000], rng.choice(idx, 20000, replace=False)])
    for i in sample:
        slots = _walk_leaf(tn[i, 0], tn[i, 1], entries, stride, runs)
        assert [int(slot_prim[k]) for k in slots] == ref_list(i), (i, slots)
        if runs and slots:
            assert slots == list(range(slots[0], slots[0] + len(slots)))
        ty = int(tn[i, 1])
        if not runs and len(slots) >= 3:
            assert ty & 0xC0000000 == 0xC0000000 and ((ty & 0x3fffffff) * 2) % 2 == 0
        if len(slots) == 0:
            assert int(tn[i, 0]) == 0xffffffff
