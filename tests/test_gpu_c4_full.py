"""BASELINE.json configs[3] at its stated scene: 10 M triangles (matte / glass / mirror mix), PathIntegrator depth 8, 2048 x 2048 film.
VERDICT r03 "missing #1": the 10 M-triangle tree (245 M nodes, depth 39; rounds 2-5: a 12 GB per-reference record array, round 6: 640 MB of records + 3.6 GB of leaf entries) had been timed but never checked.
  * rt_trace_closest / rt_trace_any of camera, random, axis-parallel, on-surface and shadow-segment rays against the CPU oracle
    walking the SAME flattened tree: hits, parameters and barycentrics bit-exact, node / leaf-reference / triangle-test counters identical;
  * the frame itself at 4 of its 256 samples per pixel (16.8 M camera samples on the full 2049 x 2049 sample extent): coverage,
    determinism, sanity, and the timed kernel's film bit-identical to its counting twin's.
The frame at all 256 spp (1.07 G camera samples, a 34 GB sample buffer) is bench.py's `c4full` workload: profiles/r04_c4_full.json.
One scene create (about half a minute on the GPU box) serves the module."""
import numpy as np
import pytest
from test_gpu_parity import need_gpu
from test_gpu_configs import accel_of, COUNTERS

pytestmark = pytest.mark.gpu

N_TRIS = 10_000_000


@pytest.fixture(scope="module")
def soup10m(pkg, scenes):
    need_gpu(pkg)
    text = scenes.cornell_scene(xres=2048, yres=2048, integrator="path", maxdepth=8, xsamples=2, ysamples=2, jitter=False, pixel_filter="box",
                                soup_tris=N_TRIS, soup_materials=True, keyed=True)
    ps = pkg.ParsedScene(text=text)
    del text
    assert ps.valid and ps.errors == 0 and ps.n_tris == N_TRIS + 12
    ds = pkg.DeviceScene(ps)
    yield ps, ds
    ds.close()


def test_10m_tree_has_the_reference_shape(soup10m):
    ps, ds = soup10m
    info = ds.accel_info()
    # KdTreeAccel: maxDepth = Round2Int(8 + 1.3 * Log2Int(N)) = 8 + 1.3 * 23 = 37.9 -> 38 (kdtree.cpp:141-147; +1 in this library's counting)
    assert info.n_tris == ps.n_tris and info.n_nodes > 200_000_000 and 38 <= info.max_depth <= 40, (info.n_nodes, info.max_depth)


def test_10m_trace_bit_exact_with_identical_work_counters(pkg, oracle, soup10m):
    ps, ds = soup10m
    nodes, refs, bounds, info = accel_of(ds)
    rng = np.random.default_rng(1010)
    n_cam = ps.n_camera_samples
    # camera rays: three bands of the frame (top, centre, bottom rows), 40 k each
    cam = np.concatenate([ds.camera_rays(first, 40_000) for first in (0, (n_cam // 2) // 4 * 4, n_cam - 40_000)])
    n = 100_000
    rnd = np.zeros(n, pkg.RAY_DTYPE)
    rnd["o"] = rng.uniform(-20, 580, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rnd["d"] = d.astype(np.float32); rnd["mint"] = 1e-3; rnd["maxt"] = np.inf
    rnd["d"][:2000] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 2000)] * rng.choice([-1, 1], (2000, 1)).astype(np.float32)
    tv = ps.tri_verts()
    rnd["o"][2000:4000] = tv[rng.integers(0, len(tv), 2000)].mean(1)        # rays that start on a triangle
    del tv
    rays = np.concatenate([cam, rnd])
    ds.reset_counters()
    hits = ds.trace_closest(rays)
    dc = ds.counters()
    ref, oc = oracle.trace(ps, rays, False, nodes, refs, bounds)
    for f in ("prim", "t", "b1", "b2"):
        assert np.array_equal(hits[f], ref[f]), f
    for k in ("nodes_visited", "leaf_refs", "tri_tests"):
        assert dc[k] == oc[k], (k, dc[k], oc[k])
    assert (hits["prim"] >= 0).mean() > 0.5
    # shadow segments from the hit points to a point on the light (VisibilityTester::SetSegment light.h:78-80)
    hit = hits["prim"] >= 0
    p = rays["o"][hit] + rays["d"][hit] * hits["t"][hit, None]
    seg = np.zeros(len(p), pkg.RAY_DTYPE)
    seg["o"] = p
    seg["d"] = (np.array([278, 548.7, 279.5], np.float32) - p).astype(np.float32)
    seg["mint"] = 1e-3; seg["maxt"] = np.float32(1.0) - np.float32(1e-3)
    ds.reset_counters()
    occ = ds.trace_any(seg)
    dc2 = ds.counters()
    refo, oc2 = oracle.trace(ps, seg, True, nodes, refs, bounds)
    assert np.array_equal(occ, refo)
    for k in ("nodes_visited", "leaf_refs", "tri_tests"):
        assert dc2[k] == oc2[k], (k, dc2[k], oc2[k])
    assert 0.02 < occ.mean() < 1.0
    # the depth-39 tree overflows the trace kernel's 8-entry LDS ring: the HBM spill path runs in earnest
    assert dc["stack_overflows"] + dc2["stack_overflows"] > 0


def test_c4_frame_properties_at_the_stated_film_size(pkg, soup10m):
    """2048 x 2048 film (2049 x 2049 sample extent), path depth 8, material mix, 4 of the 256 samples per pixel."""
    ps, ds = soup10m
    ds.set_counting(True); ds.reset_counters(); ds.bind_film(); ds.render(); a = ds.film_accum(); ca = ds.counters()
    ds.reset_counters(); ds.clear_film(); ds.render(); a2 = ds.film_accum(); ca2 = ds.counters()
    ds.set_counting(False); ds.clear_film(); ds.render(); a3 = ds.film_accum(); st = ds.last_stats()      # the timed kernel (what bench.py times on c4full)
    assert st["pipeline"] == 0
    assert ca["camera_rays"] == 2049 * 2049 * 4 and ca["bad_samples"] == 0 and ca == ca2
    assert np.array_equal(a, a2), "two renders of the counting twin differ"
    assert np.array_equal(a, a3), "the timed kernel's film differs from its counting twin's"
    assert np.all(a[4][1:-1, 1:-1] == 4.0)                                    # every camera sample exactly once (box filter, unjittered strata)
    assert np.isfinite(a).all() and a[:3].min() >= 0 and np.all(a[3] <= a[4] + 1e-3)
    assert ca["closest_rays"] >= ca["camera_rays"] and ca["any_rays"] > 0.2 * ca["camera_rays"]
    for k in COUNTERS:
        assert ca[k] > 0, k


def test_c4_sample_count_beyond_2_to_the_30_work_items_and_2_to_the_32_rays(pkg, scenes):
    """C4's stated sample count -- 2048 x 2048 film, 256 samples per pixel: 2049 * 2049 * 256 = 1 074 790 656 camera samples, just over 2^30 work items, a
    34 GB sample buffer, ~6 G rays (beyond 2^32) -- on the 1 M-triangle tree (VERDICT r04 weak #1a: the 10 M-triangle test above renders 4 of the 256
    samples per pixel, so the 32-bit work index, the sample-slot arithmetic and the 64-bit ray counters were never exercised at this size).  Box filter,
    unjittered strata: every camera sample lands in exactly one pixel, so the weight plane counts them.  Determinism (the timed kernel twice) and
    timed kernel == counting twin, bit for bit."""
    need_gpu(pkg)
    text = scenes.cornell_scene(xres=2048, yres=2048, integrator="path", maxdepth=8, xsamples=16, ysamples=16, jitter=False, pixel_filter="box",
                                soup_tris=1_000_000, soup_materials=True, keyed=True)
    ps = pkg.ParsedScene(text=text)
    del text
    assert ps.valid and ps.errors == 0 and ps.n_camera_samples == 2049 * 2049 * 256 > (1 << 30)
    ds = pkg.DeviceScene(ps)
    ds.bind_film()
    ds.set_counting(False); ds.render(); a = ds.film_accum(); st = ds.last_stats()
    assert st["pipeline"] == 0
    ds.clear_film(); ds.render(); a2 = ds.film_accum()
    assert np.array_equal(a, a2), "two renders of the timed kernel differ"
    del a2
    assert np.all(a[4][1:-1, 1:-1] == 256.0), "a camera sample was dropped or rendered twice"
    assert np.isfinite(a).all() and a[:3].min() >= 0 and np.all(a[3] <= a[4] * (1 + 1e-5))
    ds.set_counting(True); ds.reset_counters(); ds.clear_film(); ds.render(); t = ds.film_accum(); c = ds.counters()
    assert np.array_equal(a, t), "the timed kernel's film differs from its counting twin's"
    assert c["camera_rays"] == 2049 * 2049 * 256 and c["bad_samples"] == 0
    assert c["closest_rays"] + c["any_rays"] > (1 << 32) and c["closest_rays"] >= c["camera_rays"]
    assert c["nodes_visited"] > 40 * c["camera_rays"]
    ds.close()
    # the work items beyond index 2^30 (the right end of the last 32-row block row and the last sample row) are rendered like all others: the frame's
    # 64 x 64-pixel block means equal those of the same scene at 4 samples per pixel up to the noise of 4 spp
    L256 = (a[:3] / np.maximum(a[4], 1e-20)).reshape(3, 32, 64, 32, 64).mean((0, 2, 4))
    del a, t
    ps4 = pkg.ParsedScene(text=scenes.cornell_scene(xres=2048, yres=2048, integrator="path", maxdepth=8, xsamples=2, ysamples=2, jitter=False, pixel_filter="box",
                                                    soup_tris=1_000_000, soup_materials=True, keyed=True))
    d4 = pkg.DeviceScene(ps4); d4.bind_film(); d4.render(); b = d4.film_accum(); d4.close()
    L4 = (b[:3] / np.maximum(b[4], 1e-20)).reshape(3, 32, 64, 32, 64).mean((0, 2, 4))
    # (a 64 x 64 block at 4 spp is noisy where the light arrives by several bounces: rows of blocks and quadrants of the frame are compared, not single blocks)
    r256, r4 = L256.mean(1), L4.mean(1)
    assert np.all(np.abs(r256 - r4) < 0.2 * r4 + 0.02 * r4.max()), (r256.tolist(), r4.tolist())
    q256, q4 = L256.reshape(4, 8, 4, 8).mean((1, 3)), L4.reshape(4, 8, 4, 8).mean((1, 3))
    assert np.all(np.abs(q256 - q4) < 0.2 * q4 + 0.02 * q4.max()) and abs(L256.mean() - L4.mean()) < 0.05 * L4.mean(), (q256.tolist(), q4.tolist())
