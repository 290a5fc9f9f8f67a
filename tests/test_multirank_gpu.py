"""The N > 1 branch of bench.py on real hardware before an 8-GPU node is available: two ranks, launched exactly
as the driver launches them (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2), share GPU 0; the film
all-reduce goes over gloo (the backend stages device tensors through the host), everything else -- shard descriptors, the HIP render
of each rank's tiles into a full-frame film, rank-0 resolve, ray accounting, max-over-ranks timing, per-rank kernel times -- is the
code the RCCL run executes."""
import json
import os
import socket
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _bench(args, env, timeout=900, nproc=0):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args if not nproc else \
          [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("mode", [["--tile-2d", "0", "--tile-pixels", "16", "--merge", "allreduce"],      # round 1 / 2: 1-D tiles, all-reduce, rank-0 resolve
                                  ["--tile-2d", "16", "--merge", "reduce_scatter"],                         # round 3 default form: 2-D tiles, row-wise reduce-scatter,
                                  ["--tile-2d", "32", "--merge", "allreduce"]])                             # per-rank resolve, all-gather of the resolved rows
def test_two_ranks_on_one_gpu_through_bench(pkg, tmp_path, mode):
    """Two ranks share GPU 0 (collectives over gloo, staged through the host); the second rank maps the accelerator the first one
    built and published under /dev/shm (rt_scene_create_prebuilt).  Whatever the tile shape and the way the partial films are merged,
    the frame is the single-rank frame (float sums in another order: <= 2e-5 relative)."""
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible")
    env = dict(os.environ, PBRT_BENCH_BACKEND="gloo", PBRT_BENCH_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    base = ["--steps", "2", "--warmup", "1", "--workload", "tsmall", "--no-cpu-baseline", "--no-extra"] + mode
    j1 = _bench(["--gpus", "1", "--dump-film", one] + base, env)
    j2 = _bench(["--gpus", "2", "--dump-film", two] + base, env, nproc=2)
    assert j2["n_gpus"] == 2 and len(j2["per_rank"]) == 2
    # every camera sample rendered exactly once across the two shards, same rays as the single-rank frame
    assert j2["config"]["camera_samples_per_frame"] == j1["config"]["camera_samples_per_frame"]
    assert abs(j2["config"]["rays_per_frame"] - j1["config"]["rays_per_frame"]) <= 2e-4 * j1["config"]["rays_per_frame"] + 8
    assert all(r["rays"] > 0.3 * j1["config"]["rays_per_frame"] for r in j2["per_rank"])          # interleaved tiles: balanced shards
    a, b = np.load(one), np.load(two)
    assert np.allclose(a["rgb"], b["rgb"], rtol=2e-5, atol=2e-6) and np.allclose(a["alpha"], b["alpha"], rtol=2e-5, atol=2e-6)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("pbrt_hip_accel_")]            # the published tree is removed again


def test_a_failing_sub_workload_does_not_cost_the_headline(pkg):
    """bench.py at N > 1 runs the multi-GPU configurations after the headline and prints ONE line at the end: whatever goes wrong in a sub-workload (here: a
    name that does not exist; on a node: memory, a collective timing out) is recorded in its place, all ranks agree to stop, the headline line is printed."""
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible")
    env = dict(os.environ, PBRT_BENCH_BACKEND="gloo", PBRT_BENCH_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    j = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "tsmall", "--multi-workloads", "t5,nosuchworkload,t8", "--no-cpu-baseline", "--tile-2d", "16"], env, nproc=2)
    assert j["n_gpus"] == 2 and j["value"] > 0
    w = j["workloads"]
    assert [x["workload"] for x in w] == ["t5", "nosuchworkload"] and w[0]["value"] > 0 and len(w[0]["per_rank"]) == 2 and "unknown workload" in w[1]["error"]


def test_single_rank_rccl_process_group(pkg, tmp_path):
    """bench.py --force-dist: world size 1 over the nccl (= RCCL) backend on the one GPU this box has -- process-group set-up with
    device_id, the probe all-reduce, the shared-accelerator hand-shake, reduce-scatter / all-gather of device tensors through RCCL and
    the per-rank resolve all execute; the frame is the plain single-process frame."""
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    base = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "tsmall", "--no-cpu-baseline", "--no-extra"]
    j1 = _bench(base + ["--dump-film", one], env)
    for merge in ("reduce_scatter", "allreduce"):
        j2 = _bench(base + ["--dump-film", two, "--force-dist", "--merge", merge], env)
        assert j2["config"]["rays_per_frame"] == j1["config"]["rays_per_frame"]
        a, b = np.load(one), np.load(two)
        assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["alpha"], b["alpha"]), merge


def test_prebuilt_accelerator_gives_the_same_scene(pkg, scenes, tmp_path):
    """rt_scene_create_prebuilt with the arrays of another scene's tree (through a file, as bench.py's ranks hand it over): same hits,
    same film; a tree with an index out of range is refused."""
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible")
    for accel in ("kdtree", "grid"):
        ps = pkg.ParsedScene(text=scenes.cornell_scene(xres=48, yres=40, integrator="path", maxdepth=4, xsamples=2, ysamples=2, jitter=True, soup_tris=4000,
                                                       accelerator=accel, keyed=True))
        a = pkg.DeviceScene(ps); a.render(); fa = a.film_accum(); ca = a.counters()
        path = str(tmp_path / ("accel_%s.bin" % accel))
        pkg.publish_accel(a, path)
        nodes, refs, info = pkg.attach_accel(path)
        b = pkg.DeviceScene(ps, prebuilt=(nodes, refs, info)); b.render(); fb = b.film_accum(); cb = b.counters()
        assert np.array_equal(fa, fb) and ca == cb, accel
        n2, r2 = b.accel_arrays()
        assert np.array_equal(n2, nodes) and np.array_equal(r2, refs) and b.accel_info().max_depth == info.max_depth
        a.close(); b.close()
        bad = np.array(nodes, copy=True)
        if accel == "kdtree":
            interior = np.nonzero((bad[:, 0] & 3) != 3)[0]
            bad[interior[len(interior) // 2], 1] = len(bad) + 5            # an above-child beyond the array
        else:
            bad[3, 1] = len(refs) + 7                                         # a voxel list beyond the reference array
        with pytest.raises(pkg.RtError):
            pkg.DeviceScene(ps, prebuilt=(bad, refs, info))
        # the claims the device sizes scratch from are checked, not trusted: a tree deeper than its max_depth says (the
        # spill area of the traversal stack is sized from it), bounds / voxel widths that are not finite
        dup = lambda i: type(i).from_buffer_copy(bytes(i))
        if accel == "kdtree":
            shallow = dup(info); shallow.max_depth = 3
            with pytest.raises(pkg.RtError, match="deeper"):
                pkg.DeviceScene(ps, prebuilt=(nodes, refs, shallow))
        else:
            nanw = dup(info); nanw.grid_inv_width[1] = float("nan")
            with pytest.raises(pkg.RtError, match="voxel widths"):
                pkg.DeviceScene(ps, prebuilt=(nodes, refs, nanw))
        nanb = dup(info); nanb.bounds[4] = float("inf")
        with pytest.raises(pkg.RtError, match="bounds"):
            pkg.DeviceScene(ps, prebuilt=(nodes, refs, nanb))


def test_eight_ranks_on_one_gpu_rehearsal(pkg, tmp_path):
    """The shape of the driver's 8-GPU run, rehearsed on the one GPU this box has: 8 ranks launched by
    torch.distributed.run share GPU 0 over gloo; local rank 0 builds the accelerator and publishes it under /dev/shm, the other seven attach it
    (rt_scene_create_prebuilt); 2-D tiles dealt round-robin; the merge is rt_film_pack_parts + ONE reduce-scatter (a film height that is NOT a
    multiple of the world size: 426 rows, 54 per rank, the last rank's part padded) + per-rank RGBA resolve + ONE all-gather.  The run prints the
    headline record AND, as at N > 1 in production (--multi-workloads), a C4-shaped (material mix, path depth 8) and a C5-shaped (medium, march
    kernel) sub-record with per-rank figures.  Every frame is the single-rank frame, every camera sample is rendered exactly once, and the tiles
    balance the ranks: max / mean of the per-rank ray counts <= 1.05."""
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible")
    env = dict(os.environ, PBRT_BENCH_BACKEND="gloo", PBRT_BENCH_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one, eight = str(tmp_path / "one.npz"), str(tmp_path / "eight.npz")
    base = ["--steps", "1", "--warmup", "0", "--workload", "tsmall", "--multi-workloads", "t8,t5", "--no-cpu-baseline", "--tile-2d", "16", "--merge", "reduce_scatter"]
    j8 = _bench(["--gpus", "8", "--dump-film", eight] + base, env, nproc=8, timeout=2400)
    assert j8["n_gpus"] == 8 and len(j8["per_rank"]) == 8
    assert [w["workload"] for w in j8["workloads"]] == ["t8", "t5"]
    recs = {"tsmall": j8, "t8": j8["workloads"][0], "t5": j8["workloads"][1]}
    for wl, rec in recs.items():
        j1 = _bench(["--gpus", "1", "--dump-film", one, "--steps", "1", "--warmup", "0", "--workload", wl, "--no-cpu-baseline", "--no-extra", "--tile-2d", "16"], env)
        assert rec["n_gpus"] == 8 and len(rec["per_rank"]) == 8, wl
        assert rec["config"]["camera_samples_per_frame"] == j1["config"]["camera_samples_per_frame"], wl
        if wl == "t8":
            assert j1["config"]["camera_samples_per_frame"] == 644 * 430 * 4      # mitchell 2 x 2: the sample extent reaches 2 pixels beyond the film
        assert abs(rec["config"]["rays_per_frame"] - j1["config"]["rays_per_frame"]) <= 2e-4 * j1["config"]["rays_per_frame"] + 8, wl
        rays = np.array([r["rays"] for r in rec["per_rank"]], np.float64)
        assert rays.min() > 0 and rays.max() / rays.mean() <= (1.05 if wl != "tsmall" else 1.25), (wl, rays)      # (tsmall: 160 x 120 pixels are 80 tiles for 8 ranks)
        a, b = np.load(one), np.load(eight if wl == "tsmall" else eight.replace(".npz", "_%s.npz" % wl))
        if wl == "t8":
            assert a["rgb"].shape == (426, 640, 3)
        assert np.allclose(a["rgb"], b["rgb"], rtol=2e-5, atol=2e-6) and np.allclose(a["alpha"], b["alpha"], rtol=2e-5, atol=2e-6), wl
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("pbrt_hip_accel_")]
