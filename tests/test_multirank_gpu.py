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


def test_two_ranks_on_one_gpu_through_bench(pkg, tmp_path):
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible")
    env = dict(os.environ, PBRT_BENCH_BACKEND="gloo", PBRT_BENCH_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    base = ["--steps", "2", "--warmup", "1", "--workload", "tsmall", "--no-cpu-baseline", "--no-extra", "--tile-pixels", "16"]
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump-film", one] + base, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    j1 = json.loads(r1.stdout.strip().splitlines()[-1])
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                         os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dump-film", two] + base, env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, (r2.stdout[-1500:], r2.stderr[-3000:])
    j2 = json.loads([ln for ln in r2.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and len(j2["per_rank"]) == 2
    # every camera sample rendered exactly once across the two shards, same rays as the single-rank frame
    assert j2["config"]["camera_samples_per_frame"] == j1["config"]["camera_samples_per_frame"]
    assert abs(j2["config"]["rays_per_frame"] - j1["config"]["rays_per_frame"]) <= 2e-4 * j1["config"]["rays_per_frame"] + 8
    assert all(r["rays"] > 0.3 * j1["config"]["rays_per_frame"] for r in j2["per_rank"])          # interleaved tiles: balanced shards
    a, b = np.load(one), np.load(two)
    assert np.allclose(a["rgb"], b["rgb"], rtol=2e-5, atol=2e-6) and np.allclose(a["alpha"], b["alpha"], rtol=2e-5, atol=2e-6)
