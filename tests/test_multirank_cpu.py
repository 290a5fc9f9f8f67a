"""The N > 1 path on CPU: world_size 2 over gloo.  Each rank renders its round-robin tile shard (the partition the
host front end hands to rt_render; here evaluated by the CPU oracle) into a full-frame 5-plane film, one
all-reduce(sum) merges them -- exactly bench.py's step with backend nccl (= RCCL) swapped for gloo.
The merged film must equal the single-rank film: tile/shard logic and keyed RNG are partition invariant."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, scene_text, tile, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    import oracle
    pkg = entry.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps = pkg.ParsedScene(text=scene_text)
    ps.set_shard(rank, world, tile)
    nodes, refs, bounds, _ = ps.kdtree()
    _, _, accum, cnt = oracle.render(ps, nodes, refs, bounds)
    film = torch.from_numpy(accum.copy())
    dist.all_reduce(film, op=dist.ReduceOp.SUM)
    rays = torch.tensor([cnt["camera_rays"], cnt["closest_rays"], cnt["any_rays"]], dtype=torch.int64)
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(os.path.join(out_dir, "film.npy"), film.numpy())
        np.save(os.path.join(out_dir, "rays.npy"), rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


def pack_parts(accum, world, rows):
    """numpy restatement of rt_film_pack_parts (include/pbrt_hip.h): [5][H][W] -> [world][5][rows][W], rows beyond H zero"""
    _, H, W = accum.shape
    pad = np.zeros((5, world * rows, W), np.float32)
    pad[:, :H] = accum
    return np.ascontiguousarray(pad.reshape(5, world, rows, W).transpose(1, 0, 2, 3))


def _worker_parts(rank, world, port, scene_text, tile, out_dir):
    """bench.py's N > 1 step since round 5: pack, ONE reduce-scatter, per-rank resolve of the rank's rows, ONE all-gather of RGBA rows"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    import oracle
    pkg = entry.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps = pkg.ParsedScene(text=scene_text)
    ps.set_shard(rank, world, tile)
    nodes, refs, bounds, _ = ps.kdtree()
    _, _, accum, _ = oracle.render(ps, nodes, refs, bounds)
    H, W = accum.shape[1:]
    rows = (H + world - 1) // world
    part = torch.zeros((5, rows, W))
    dist.reduce_scatter_tensor(part.view(-1), torch.from_numpy(pack_parts(accum, world, rows)).view(-1))
    rgb, alpha = oracle.resolve(part.numpy().copy(), premultiply=ps.premultiply)
    rgba_all = torch.zeros((world * rows, W, 4))
    dist.all_gather_into_tensor(rgba_all, torch.from_numpy(np.concatenate([rgb, alpha[..., None]], axis=-1).astype(np.float32)))
    if rank == 1:                                                     # (any rank holds the whole frame)
        np.save(os.path.join(out_dir, "rgba.npy"), rgba_all.numpy()[:H])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tile", [(8, 8), 7])
def test_two_rank_fused_merge_gives_the_single_rank_image(pkg, scenes, oracle, tmp_path, tile):
    """21 film rows over 2 ranks: 11 each, the second rank's part padded by one row of zeros."""
    import torch.multiprocessing as mp
    text = scenes.cornell_scene(xres=24, yres=21, integrator="path", xsamples=2, ysamples=2, jitter=True, pixel_filter="mitchell", keyed=True)
    ps = pkg.ParsedScene(text=text)
    nodes, refs, bounds, _ = ps.kdtree()
    rgb, alpha, _, _ = oracle.render(ps, nodes, refs, bounds)
    port = 31500 + (os.getpid() + (tile if isinstance(tile, int) else 100 + tile[0])) % 2000
    mp.spawn(_worker_parts, args=(2, port, text, tile, str(tmp_path)), nprocs=2, join=True)
    rgba = np.load(tmp_path / "rgba.npy")
    assert rgba.shape == (21, 24, 4)
    assert np.allclose(rgba[..., :3], rgb, rtol=2e-5, atol=2e-6) and np.allclose(rgba[..., 3], alpha, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("tile", [7, 64, (8, 8), (16, 5)])
def test_two_rank_tile_shards_sum_to_the_full_film(pkg, scenes, oracle, tmp_path, tile):
    import torch.multiprocessing as mp
    text = scenes.cornell_scene(xres=24, yres=20, integrator="path", xsamples=2, ysamples=2, jitter=True,
                                pixel_filter="mitchell", keyed=True)
    ps = pkg.ParsedScene(text=text)
    nodes, refs, bounds, _ = ps.kdtree()
    _, _, full, cnt = oracle.render(ps, nodes, refs, bounds)
    port = 29500 + (os.getpid() + (tile if isinstance(tile, int) else 100 + tile[0] + tile[1])) % 2000
    mp.spawn(_worker, args=(2, port, text, tile, str(tmp_path)), nprocs=2, join=True)
    merged = np.load(tmp_path / "film.npy")
    rays = np.load(tmp_path / "rays.npy")
    assert rays.tolist() == [cnt["camera_rays"], cnt["closest_rays"], cnt["any_rays"]]
    # every sample is evaluated by exactly one rank with the same key; only the order of float additions into a
    # pixel differs between (a+b)+(c+d) and ((a+b)+c)+d
    assert np.allclose(merged, full, rtol=2e-6, atol=1e-6)
    assert np.array_equal(merged[4] != 0, full[4] != 0)


def test_shard_partition_covers_every_sample_once(pkg, scenes, oracle):
    text = scenes.cornell_scene(xres=17, yres=9, integrator="whitted", xsamples=2, ysamples=1, keyed=True)
    total = None
    for world in (1, 3, 5):
        cams = 0
        for r in range(world):
            ps = pkg.ParsedScene(text=text); ps.set_shard(r, world, 4)
            cams += oracle.render(ps)[3]["camera_rays"]
        total = cams if total is None else total
        assert cams == total == 18 * 10 * 2
        for tile in ((4, 4), (5, 3), (32, 32)):                      # 2-D tiles, clipped at the border of the 18 x 10 sample extent
            cams = 0
            for r in range(world):
                ps = pkg.ParsedScene(text=text); ps.set_shard(r, world, tile)
                cams += oracle.render(ps)[3]["camera_rays"]
            assert cams == total, (world, tile)


def test_fitted_tiles_pad_the_extent_least(pkg, scenes):
    """ParsedScene.set_shard(fit=True): 2-D tile sizes within 3/4 .. 5/4 of the request whose whole tiles overshoot the sample extent least
    (a padded work item idles a lane for about a ray's time on the device; 64 x 64 tiles pad a 1025 x 1025 extent by 12.7 %)."""
    for extent, size in ((1921, 64), (1081, 64), (1025, 64), (1028, 64), (37, 16), (5, 64), (4096, 64)):
        t = pkg.fit_tile(extent, size)
        assert max(1, size * 3 // 4) <= t <= max(2, size * 5 // 4)
        pad = -(-extent // t) * t - extent
        for other in range(max(1, size * 3 // 4), max(2, size * 5 // 4) + 1):
            assert pad <= -(-extent // other) * other - extent
    assert pkg.fit_tile(4096, 64) == 64 and pkg.fit_tile(1025, 64) in (41, 57)           # 25 x 41 and 18 x 57 - 1: both overshoot by at most 1
    ps = pkg.ParsedScene(text=scenes.cornell_scene(xres=1024, yres=1024, pixel_filter="box"))
    x0, x1, y0, y1 = ps.sample_extent
    w, h = ps.set_shard(1, 4, (64, 64), fit=True)
    assert (w, h) == (pkg.fit_tile(x1 - x0, 64), pkg.fit_tile(y1 - y0, 64))
    assert ps.set_shard(1, 4, (64, 64)) == (64, 64) and ps.set_shard(0, 1, 48) == 48
