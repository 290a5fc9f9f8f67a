"""GPU tests of the BASELINE.json configs the golden fixtures are too small for:
  C3  1 M-triangle soup, DirectLighting, 1920x1080 @ 16 spp    -> full-size properties + oracle parity of a crop-sized frame
  C4  matte / glass / mirror soup mix, PathIntegrator maxdepth 8 -> device vs the CPU oracle at 30 k triangles
  C5  homogeneous medium + single scattering over a soup        -> device vs the CPU oracle at 20 k triangles
  and the >= 1 M-triangle regime itself (depth-34 tree, 36 M nodes, HBM stack spills): rt_trace_closest / rt_trace_any of
  camera, random and shadow-segment rays bit-exact against the oracle on the SAME tree, with identical work counters.
Everything goes through the C ABI (pbrt_hip.h); the oracle is the checker only."""
import numpy as np
import pytest
from conftest import film_metrics
from test_gpu_parity import need_gpu, check_film, record_case

pytestmark = pytest.mark.gpu

COUNTERS = ("camera_rays", "closest_rays", "any_rays", "nodes_visited", "leaf_refs", "tri_tests")


def accel_of(ds):
    nodes, refs = ds.accel_arrays()
    info = ds.accel_info()
    return nodes, refs, np.array(list(info.bounds), np.float32), info


# ---------------------------------------------------------------------------------------------- C4 / C5 at oracle size
@pytest.mark.parametrize("cfg", [
    # C4's workload: material mix (tri % 10: glass, mirror, matte) + path tracing to depth 8
    dict(xres=128, yres=128, integrator="path", maxdepth=8, xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell",
         soup_tris=30000, soup_materials=True),
    dict(xres=96, yres=96, integrator="path", maxdepth=8, sampler="lowdiscrepancy", pixelsamples=4, soup_tris=50000, soup_materials=True),
    # C5's workload: homogeneous medium, single scattering, over a soup (with both surface integrators the config can mean)
    dict(xres=96, yres=96, integrator="directlighting", xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell", soup_tris=20000,
         volume_integrator='"single" "float stepsize" [40]', world_kwargs=dict(volume='"float g" [.2]')),
    dict(xres=80, yres=80, integrator="path", maxdepth=5, xsamples=2, ysamples=2, jitter=True, soup_tris=20000, soup_materials=True,
         volume_integrator='"single" "float stepsize" [60]', world_kwargs=dict(volume='"float g" [-.1] "color Le" [.001 .001 .002]')),
    dict(xres=96, yres=96, integrator="whitted", xsamples=2, ysamples=1, soup_tris=25000, soup_materials=True,
         volume_integrator='"emission" "float stepsize" [30]', world_kwargs=dict(volume='"color Le" [.002 .001 .001]')),
])
def test_c4_c5_workloads_against_the_oracle(pkg, scenes, oracle, cfg):
    need_gpu(pkg)
    ps = pkg.ParsedScene(text=scenes.cornell_scene(keyed=True, **cfg))
    assert ps.valid and ps.errors == 0 and ps.warnings == 0
    ds = pkg.DeviceScene(ps)
    ds.render()
    rgb, alpha = ds.film()
    cnt = ds.counters()
    nodes, refs, bounds, info = accel_of(ds)
    # the timed flavours give the same film as the counting twin just compared
    acc = ds.film_accum()
    for env in (dict(PBRT_HIP_HIGH_OCC="0"), dict(PBRT_HIP_HIGH_OCC="1"), dict(PBRT_HIP_PIPELINE="1")):
        with pytest.MonkeyPatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            ds.set_counting(False); ds.clear_film(); ds.render()
            assert np.array_equal(ds.film_accum(), acc), env
    ds.close()
    orgb, oalpha, _, ocnt = oracle.render(ps, nodes, refs, bounds, info=info)
    m = check_film("cfg:" + str(cfg), rgb, alpha, orgb, oalpha, ps.integrator)
    record_case("c4c5:%s:%d" % (cfg["integrator"], cfg["soup_tris"]), m)
    assert cnt["camera_rays"] == ocnt["camera_rays"] and cnt["bad_samples"] == 0
    if ps.integrator != 2:
        for k in COUNTERS:
            assert cnt[k] == ocnt[k], (k, cnt[k], ocnt[k])
    else:
        for k in ("closest_rays", "any_rays", "nodes_visited", "tri_tests"):
            assert abs(cnt[k] - ocnt[k]) <= 5e-4 * ocnt[k] + 8, (k, cnt[k], ocnt[k])


# ---------------------------------------------------------------------------------------------- the 1 M-triangle regime
@pytest.fixture(scope="module")
def soup1m(pkg, scenes):
    """C3's scene at a frame the oracle finishes in seconds; the tree is the full 36 M-node, depth-34 one."""
    need_gpu(pkg)
    text = scenes.cornell_scene(xres=96, yres=54, integrator="directlighting", xsamples=2, ysamples=2, jitter=True,
                                pixel_filter="mitchell", soup_tris=1_000_000, keyed=True)
    ps = pkg.ParsedScene(text=text)
    assert ps.valid and ps.errors == 0 and ps.n_tris == 1_000_012
    ds = pkg.DeviceScene(ps)
    yield ps, ds, accel_of(ds)
    ds.close()


def test_1m_tree_is_the_reference_sized_tree(soup1m):
    ps, ds, (nodes, refs, bounds, info) = soup1m
    # maxDepth = Round2Int(8 + 1.3 * Log2Int(N)) = 8 + 1.3 * 19 = 32.7 -> 33 (+1 counting convention here)
    assert info.n_nodes > 30_000_000 and info.max_depth >= 33 and info.n_tris == ps.n_tris


def test_1m_trace_bit_exact_with_identical_work_counters(pkg, oracle, soup1m):
    ps, ds, (nodes, refs, bounds, info) = soup1m
    rng = np.random.default_rng(77)
    cam = ds.camera_rays(0, ps.n_camera_samples)                              # every camera ray of the frame
    n = 120_000
    rnd = np.zeros(n, pkg.RAY_DTYPE)
    rnd["o"] = rng.uniform(-20, 580, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rnd["d"] = d.astype(np.float32); rnd["mint"] = 1e-3; rnd["maxt"] = np.inf
    rnd["d"][:2000] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 2000)] * rng.choice([-1, 1], (2000, 1)).astype(np.float32)
    tv = ps.tri_verts()
    rnd["o"][2000:4000] = tv[rng.integers(0, len(tv), 2000)].mean(1)        # rays that start on a triangle
    rays = np.concatenate([cam, rnd])
    ds.reset_counters()
    hits = ds.trace_closest(rays)
    dc = ds.counters()
    ref, oc = oracle.trace(ps, rays, False, nodes, refs, bounds)
    assert np.array_equal(hits["prim"], ref["prim"]) and np.array_equal(hits["t"], ref["t"])
    assert np.array_equal(hits["b1"], ref["b1"]) and np.array_equal(hits["b2"], ref["b2"])
    for k in ("nodes_visited", "leaf_refs", "tri_tests"):
        assert dc[k] == oc[k], (k, dc[k], oc[k])
    assert (hits["prim"] >= 0).mean() > 0.5
    # shadow segments from the hit points towards the light (VisibilityTester::SetSegment light.h:78-80): the longest rays
    hit = hits["prim"] >= 0
    p = rays["o"][hit] + rays["d"][hit] * hits["t"][hit, None]
    seg = np.zeros(len(p), pkg.RAY_DTYPE)
    seg["o"] = p
    seg["d"] = (np.array([278, 548.7, 279.5], np.float32) - p).astype(np.float32)
    seg["mint"] = 1e-3; seg["maxt"] = np.float32(1.0) - np.float32(1e-3)
    ds.reset_counters()
    occ = ds.trace_any(seg)
    dc = ds.counters()
    refo, oc = oracle.trace(ps, seg, True, nodes, refs, bounds)
    assert np.array_equal(occ, refo)
    for k in ("nodes_visited", "leaf_refs", "tri_tests"):
        assert dc[k] == oc[k], (k, dc[k], oc[k])
    # a depth-34 tree overflows the trace kernel's 8-entry LDS ring: the spill path to HBM is exercised in earnest here
    assert dc["stack_overflows"] > 0, dc


def test_1m_direct_lighting_frame_against_the_oracle(pkg, oracle, soup1m, monkeypatch):
    """A C3 frame (reduced resolution) rendered by every kernel flavour: the counting twin against the oracle (bit-exact film,
    identical counters), then the timed flavours (3 waves/SIMD, 4 waves/SIMD, the queue pipeline) against the twin."""
    ps, ds, (nodes, refs, bounds, info) = soup1m
    ds.set_counting(True); ds.reset_counters(); ds.clear_film() if ds._film_bound else None
    ds.render()
    rgb, alpha = ds.film()
    acc = ds.film_accum()
    cnt = ds.counters()
    orgb, oalpha, _, ocnt = oracle.render(ps, nodes, refs, bounds, info=info)
    check_film("c3_1m", rgb, alpha, orgb, oalpha, ps.integrator)
    for k in COUNTERS:
        assert cnt[k] == ocnt[k], (k, cnt[k], ocnt[k])
    assert cnt["stack_overflows"] > 0                  # the 12-entry ring does spill on this tree
    for env in (dict(PBRT_HIP_HIGH_OCC="0"), dict(PBRT_HIP_HIGH_OCC="1"), dict(PBRT_HIP_PIPELINE="1"), dict(PBRT_HIP_PIPELINE="0")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ds.set_counting(False); ds.clear_film(); ds.render()
        assert np.array_equal(ds.film_accum(), acc), env
        for k in env:
            monkeypatch.delenv(k)
    ds.set_counting(True)


def test_1m_path_frame_against_the_oracle(pkg, scenes, oracle):
    """The north star's case at a frame the oracle finishes in seconds: PathIntegrator depth 5 on the full 36 M-node tree (64x36 @ 4).
    Counting twin against the oracle (the path bar: >= 99.5 % of pixels with L2 < 1e-4; ray counts within 5e-4), then every timed
    flavour -- the megakernel at 4 waves per SIMD (the default for a frame without a medium) and at 3, the queue pipeline by vertex and per ray --
    bit-identical to the twin."""
    need_gpu(pkg)
    ps = pkg.ParsedScene(text=scenes.cornell_scene(xres=64, yres=36, integrator="path", maxdepth=5, xsamples=2, ysamples=2, jitter=True,
                                                   pixel_filter="mitchell", soup_tris=1_000_000, keyed=True))
    assert ps.valid and ps.errors == 0 and ps.n_tris == 1_000_012
    ds = pkg.DeviceScene(ps)
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("PBRT_HIP_PIPELINE", "1")
        ds.render()
    assert ds.last_stats()["pipeline"] == 1                      # the counting twin of the by-vertex pipeline
    rgb, alpha = ds.film(); acc = ds.film_accum(); cnt = ds.counters()
    nodes, refs, bounds, info = accel_of(ds)
    orgb, oalpha, _, ocnt = oracle.render(ps, nodes, refs, bounds, info=info)
    m = check_film("p1m_path", rgb, alpha, orgb, oalpha, ps.integrator)
    record_case("p1m:path", m)
    assert m["frac"] >= 0.995, m
    assert cnt["camera_rays"] == ocnt["camera_rays"] and cnt["bad_samples"] == 0
    for k in ("closest_rays", "any_rays", "nodes_visited", "tri_tests"):
        assert abs(cnt[k] - ocnt[k]) <= 5e-4 * ocnt[k] + 8, (k, cnt[k], ocnt[k])
    for env in (dict(PBRT_HIP_PIPELINE="1"), dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_PIPE_VERTEX="0"), dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_PIPE_SLOTS="1024"),
                dict(PBRT_HIP_PIPELINE="0", PBRT_HIP_HIGH_OCC="0"), dict(PBRT_HIP_PIPELINE="0", PBRT_HIP_HIGH_OCC="1")):
        with pytest.MonkeyPatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            ds.set_counting(False); ds.clear_film(); ds.render()
            assert np.array_equal(ds.film_accum(), acc), env
    # the counting twin of the megakernel sees the same rays and does the same traversal work
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("PBRT_HIP_PIPELINE", "0")
        ds.set_counting(True); ds.reset_counters(); ds.clear_film(); ds.render()
        c2 = ds.counters()
        assert np.array_equal(ds.film_accum(), acc)
        for k in ("camera_rays", "closest_rays", "any_rays", "nodes_visited", "leaf_refs", "tri_tests", "bad_samples"):
            assert c2[k] == cnt[k], (k, c2[k], cnt[k])
    ds.close()


@pytest.mark.parametrize("name", ["p1m", "c4", "c5"])
def test_pipeline_workloads_full_size_properties(pkg, scenes, name):
    """The three pipeline workloads bench.py times, at the size it times them (1 M-triangle soup in the Cornell box, 1024x1024):
    the path frame (depth 5, 16 spp), C4's material mix (path depth 8, 16 spp), C5's medium (single scattering, stepsize 20, g 0 +
    DirectLighting; all of its 64 spp).  No oracle at this size:
      coverage     every camera sample rendered exactly once: box filter, unjittered strata -> weight spp on every interior pixel;
      determinism  two renders of the counting twin give the bit-identical film and counters;
      flavours     the timed pipeline (what bench.py times) and the timed megakernel give that same film;
      sanity       no NaN / negative / infinite sample, alpha <= weight;
      linearity    (path frame) doubling the emitter's L doubles every radiance accumulator and changes no ray count."""
    need_gpu(pkg)
    kw = dict(xres=1024, yres=1024, xsamples=4, ysamples=4, jitter=False, pixel_filter="box", soup_tris=1_000_000, keyed=True)
    if name == "p1m": kw.update(integrator="path", maxdepth=5)
    elif name == "c4": kw.update(integrator="path", maxdepth=8, soup_materials=True)
    else: kw.update(integrator="directlighting", xsamples=8, ysamples=8, volume_integrator='"single" "float stepsize" [20]', world_kwargs=dict(volume='"float g" [0]'))
    spp = kw["xsamples"] * kw["ysamples"]
    ps = pkg.ParsedScene(text=scenes.cornell_scene(**kw))
    assert ps.valid and ps.errors == 0
    ds = pkg.DeviceScene(ps)
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("PBRT_HIP_PIPELINE", "1")
        ds.render(); a = ds.film_accum(); ca = ds.counters(); st = ds.last_stats()
        assert st["pipeline"] == 1
        ds.reset_counters(); ds.clear_film(); ds.render(); a2 = ds.film_accum(); ca2 = ds.counters()
        ds.set_counting(False); ds.clear_film(); ds.render(); a3 = ds.film_accum()       # the timed pipeline
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("PBRT_HIP_PIPELINE", "0")
        ds.clear_film(); ds.render(); a4 = ds.film_accum()                               # the timed megakernel
        assert ds.last_stats()["pipeline"] == 0
    ds.close()
    assert ca["camera_rays"] == 1025 * 1025 * spp and ca["bad_samples"] == 0 and ca == ca2
    assert np.array_equal(a, a2) and np.array_equal(a, a3) and np.array_equal(a, a4)
    assert np.all(a[4][1:-1, 1:-1] == float(spp))
    assert np.isfinite(a).all() and a[:3].min() >= 0 and np.all(a[3] <= a[4] + 1e-3)
    assert ca["closest_rays"] >= ca["camera_rays"] and ca["any_rays"] > 0.2 * ca["camera_rays"]
    if name == "p1m":
        kw["world_kwargs"] = dict(light_L=(34, 24, 8))
        ps2 = pkg.ParsedScene(text=scenes.cornell_scene(**kw))
        ds2 = pkg.DeviceScene(ps2); ds2.render(); b = ds2.film_accum(); cb = ds2.counters(); ds2.close()
        for k in COUNTERS:
            assert cb[k] == ca[k], k
        assert np.allclose(b[:3], 2 * a[:3], rtol=1e-5, atol=1e-5) and np.array_equal(b[4], a[4]) and np.array_equal(b[3], a[3])


def test_c3_full_size_properties(pkg, scenes):
    """BASELINE configs[2] at full size: 1 M triangles, DirectLighting, 1920x1080 @ 16 spp.  No oracle at this size:
      coverage     every camera sample rendered exactly once (box filter, unjittered 4x4 strata: weight 16 per interior pixel);
      determinism  two renders give the bit-identical film and counters;
      sanity       no NaN / negative / infinite sample, alpha <= weight;
      linearity    doubling the emitter's L doubles every radiance accumulator and changes no ray count."""
    need_gpu(pkg)
    kw = dict(xres=1920, yres=1080, integrator="directlighting", xsamples=4, ysamples=4, jitter=False, pixel_filter="box",
              soup_tris=1_000_000, keyed=True)
    ps = pkg.ParsedScene(text=scenes.cornell_scene(**kw))
    assert ps.valid and ps.errors == 0
    ds = pkg.DeviceScene(ps); ds.render(); a = ds.film_accum(); ca = ds.counters()
    ds.reset_counters(); ds.clear_film(); ds.render(); a2 = ds.film_accum(); ca2 = ds.counters()
    ds.set_counting(False); ds.clear_film(); ds.render(); a3 = ds.film_accum()       # the timed flavour the bench uses
    ds.close()
    assert ca["camera_rays"] == 1921 * 1081 * 16 and ca["bad_samples"] == 0 and ca == ca2
    assert np.array_equal(a, a2) and np.array_equal(a, a3)
    assert np.all(a[4][1:-1, 1:-1] == 16.0)
    assert np.isfinite(a).all() and a[:3].min() >= 0 and np.all(a[3] <= a[4] + 1e-3)
    assert ca["closest_rays"] >= ca["camera_rays"] and ca["any_rays"] > 0.2 * ca["camera_rays"]
    ps2 = pkg.ParsedScene(text=scenes.cornell_scene(world_kwargs=dict(light_L=(34, 24, 8)), **kw))
    ds2 = pkg.DeviceScene(ps2); ds2.render(); b = ds2.film_accum(); cb = ds2.counters(); ds2.close()
    for k in COUNTERS:
        assert cb[k] == ca[k], k
    assert np.allclose(b[:3], 2 * a[:3], rtol=1e-5, atol=1e-5) and np.array_equal(b[4], a[4]) and np.array_equal(b[3], a[3])


# ---------------------------------------------------------------------------------------------- per-vertex N / uv / S
@pytest.mark.parametrize("integrator,extra", [("whitted", dict(xsamples=2, ysamples=1)), ("directlighting", dict(xsamples=2, ysamples=2, jitter=True)),
                                              ("path", dict(xsamples=2, ysamples=2, jitter=True, maxdepth=6))])
def test_smooth_meshes_against_the_oracle(pkg, scenes, oracle, integrator, extra):
    """Triangle::GetShadingGeometry (trianglemesh.cpp:71-133) at a size the fixtures do not reach: finely tessellated blobs with
    N, N + uv (mirrored mapping), N + S + uv and S only, under rotations / non-uniform scales / ReverseOrientation, every material,
    inside the Cornell box with a 4 k-triangle soup; device (EXT kernels) against the oracle, all kernel flavours bit-identical."""
    need_gpu(pkg)
    def mesh(mat, xf, **kw):
        return "AttributeBegin\n%s\n%s\n%sAttributeEnd\n" % (mat, xf, scenes.smooth_mesh_text(nu=28, nv=18, **kw))
    world = (mesh('Material "matte" "color Kd" [.7 .6 .3] "float sigma" [20]', "Translate 150 110 330\nRotate 33 1 1 0", radius=95, with_n=True, with_uv=False, squash=(1, .8, 1)) +
             mesh('Material "plastic" "color Kd" [.2 .3 .7] "float roughness" [.1]', "ReverseOrientation\nTranslate 410 130 220\nScale -1 1.3 .8", radius=75, with_n=True, with_uv=True, mirror_uv=True) +
             mesh('Material "glass" "float index" [1.45]', "Translate 300 330 300\nRotate 70 0 1 0", radius=80, with_n=True, with_uv=True, with_s=True, squash=(1, 1, .85)) +
             mesh('Material "mirror"', "Translate 120 400 200", radius=60, with_n=False, with_uv=False, with_s=True) +
             mesh('Material "uber" "color Kd" [.4 .4 .2] "color Kr" [.2 .2 .2]', "Translate 430 420 420\nRotate 15 0 0 1", radius=55, with_n=True, with_uv=True))
    ps = pkg.ParsedScene(text=scenes.cornell_scene(xres=112, yres=112, integrator=integrator, soup_tris=4000, keyed=True, seed=5,
                                                   world_kwargs=dict(extra=world, point_light=(integrator == "whitted")), **extra))
    assert ps.valid and ps.errors == 0 and ps.warnings == 0
    ds = pkg.DeviceScene(ps)
    ds.render()
    rgb, alpha = ds.film()
    cnt = ds.counters()
    acc = ds.film_accum()
    nodes, refs, bounds, info = accel_of(ds)
    for env in (dict(PBRT_HIP_PIPELINE="0"), dict(PBRT_HIP_PIPELINE="1")):
        with pytest.MonkeyPatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            ds.set_counting(False); ds.clear_film(); ds.render()
            assert np.array_equal(ds.film_accum(), acc), env
    ds.close()
    orgb, oalpha, _, ocnt = oracle.render(ps, nodes, refs, bounds, info=info)
    m = check_film("smooth:" + integrator, rgb, alpha, orgb, oalpha, 2 if integrator == "path" else ps.integrator)
    assert cnt["camera_rays"] == ocnt["camera_rays"] and cnt["bad_samples"] == 0
    if integrator != "path":
        for k in COUNTERS:
            assert cnt[k] == ocnt[k], (k, cnt[k], ocnt[k])


def test_crop_window_tiles_to_exr_and_assembled(pkg, scenes, tmp_path):
    """The reference's own multi-process mode: one `cropwindow` render per tile (film/image.cpp:220-228), each
    written as an EXR whose data window sits inside the full display window (exrio.cpp:75-96), merged by exrassemble
    (tools/exrassemble.cpp:42-75).  Device films -> rt_film_resolve -> write_exr -> assemble_exr must equal the EXR of the full
    frame (a scene no random draw reaches, so tiles and full frame take identical samples)."""
    need_gpu(pkg)
    kw = dict(xres=96, yres=64, integrator="whitted", xsamples=2, ysamples=2, jitter=False, pixel_filter="mitchell", soup_tris=2000,
              world_kwargs=dict(point_light=True, area_light=False))
    rgb, alpha, _, _ = pkg.render_text(scenes.cornell_scene(**kw))
    full = str(tmp_path / "full.exr"); pkg.write_exr(full, rgb, alpha)
    paths = []
    for k, crop in enumerate([(0, .5, 0, .5), (.5, 1, 0, .5), (0, .25, .5, 1), (.25, 1, .5, 1)]):
        ps = pkg.ParsedScene(text=scenes.cornell_scene(crop=crop, **kw))
        ds = pkg.DeviceScene(ps); ds.render(); r, a = ds.film(); ds.close()
        x0, y0 = int(np.ceil(96 * crop[0])), int(np.ceil(64 * crop[2]))
        assert r.shape[:2] == (ps.height, ps.width)
        p = str(tmp_path / ("tile%d.exr" % k)); paths.append(p)
        pkg.write_exr(p, r, a, total_res=(96, 64), offset=(x0, y0))
    out = str(tmp_path / "assembled.exr")
    assert pkg.assemble_exr(paths, out) == 1.0
    a_rgb, a_alpha, meta = pkg.read_exr(out)
    f_rgb, f_alpha, _ = pkg.read_exr(full)
    assert meta == dict(total_res=(96, 64), offset=(0, 0))
    assert np.array_equal(a_rgb, f_rgb) and np.array_equal(a_alpha, f_alpha)
    with np.errstate(over="ignore"):
        assert np.array_equal(f_rgb, rgb.astype(np.float16).astype(np.float32))


def test_pipeline_pool_is_sized_by_a_memory_budget(pkg, scenes):
    """The queue pipeline keeps its per-slot scratch (state planes, recursion frames, the ray march's LatinHypercube table: 3 floats per
    march step) for every slot of the pool; a fine stepsize times 8 M slots would ask for ~100 GB.  The pool is cut to a memory budget
    instead (rt_kernels.hip render_pipeline): same film, more iterations."""
    need_gpu(pkg)
    cfg = dict(xres=64, yres=64, integrator="directlighting", xsamples=2, ysamples=2, jitter=True, soup_tris=5000,
               volume_integrator='"single" "float stepsize" [3]', world_kwargs=dict(volume='"float g" [.1]'))
    ps = pkg.ParsedScene(text=scenes.cornell_scene(keyed=True, **cfg))
    assert ps.valid and ps.errors == 0
    ds = pkg.DeviceScene(ps)
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("PBRT_HIP_PIPELINE", "0")
        ds.set_counting(False); ds.render(); ref = ds.film_accum()
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("PBRT_HIP_PIPELINE", "1")
        ds.clear_film(); ds.render(); full = ds.film_accum(); st_full = ds.last_stats()
        mp.setenv("PBRT_HIP_PIPE_MEM_MB", "4")          # ~4.5 KB per slot at ~340 march steps -> a pool of a few hundred slots
        ds.clear_film(); ds.render(); small = ds.film_accum(); st_small = ds.last_stats()
    ds.close()
    assert st_full["pipeline"] == 1 and st_small["pipeline"] == 1
    assert st_small["slots"] < st_full["slots"] and st_small["slots"] <= 2048 and st_small["iterations"] > st_full["iterations"]
    assert np.array_equal(full, ref) and np.array_equal(small, ref)


# ---------------------------------------------------------------------------------------------- KdTreeAccel's build parameters ("report both")
@pytest.mark.parametrize("integrator", ["whitted", "directlighting", "path"])
def test_tuned_tree_gives_the_same_film(pkg, scenes, integrator):
    """bench.py's `<workload>_tuned` sub-records render the same frame on the tree other build parameters give (accelerators/kdtree.cpp:489-498; bench.TUNED_ACCEL and
    two more sets).  A closest hit is the smallest t over ALL primitives and an occlusion test is a yes / no: neither depends on how the tree is cut, and the keyed
    sample stream does not depend on the traversal -- so the film must be the default tree's film BIT FOR BIT (Whitted, DirectLighting, and the path integrator too),
    with the same rays, while the work counters differ (fewer nodes visited, more triangles tested)."""
    need_gpu(pkg)
    import bench
    kw = dict(xres=96, yres=80, integrator=integrator, maxdepth=4, xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell", soup_tris=40000,
              soup_materials=True, keyed=True)
    base = None
    for params in ("", bench.TUNED_ACCEL, '"integer intersectcost" [2] "integer traversalcost" [1] "float emptybonus" [0.2] "integer maxprims" [8]',
                   '"integer intersectcost" [20] "integer traversalcost" [3] "integer maxprims" [4] "integer maxdepth" [12]'):
        ps = pkg.ParsedScene(text=scenes.cornell_scene(accel_params=params, **kw))
        assert ps.valid and ps.errors == 0 and ps.warnings == 0, params
        ds = pkg.DeviceScene(ps)
        ds.render()
        acc, cnt, info = ds.film_accum(), ds.counters(), ds.accel_info()
        ds.set_counting(False); ds.clear_film(); ds.render()
        assert np.array_equal(ds.film_accum(), acc), params                      # the timed kernel on that tree
        ds.close()
        if base is None:
            base = (acc, cnt, info.n_nodes)
            continue
        assert info.n_nodes != base[2], params                                     # really another tree
        assert np.array_equal(acc, base[0]), params
        for k in ("camera_rays", "closest_rays", "any_rays", "bad_samples"):
            assert cnt[k] == base[1][k], (params, k)
        assert cnt["nodes_visited"] != base[1]["nodes_visited"]
