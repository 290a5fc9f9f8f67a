import glob
import json
import os
import sys

import numpy as np
import pytest

os.environ.setdefault("PBRT_HIP_TUNE", "1")      # the library reads its PBRT_HIP_* experiment / flavour knobs only then (rt_kernels.hip knob())

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as entry
    entry.build()
    return entry.load_package()


@pytest.fixture(scope="session")
def scenes(pkg):
    from pbrt_v1_amd import scenes as s
    return s


@pytest.fixture(scope="session")
def oracle(pkg):
    import oracle as o       # oracle/oracle.py -- the checker; tests only
    o.lib()
    return o


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz"))
                  if not os.path.basename(p).startswith("probe_") or prefix == "probe_")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["scene"] = str(d["scene"])
    d["stats"] = json.loads(str(d["stats"]))
    return d


def film_metrics(rgb, ref):
    d = rgb.astype(np.float64) - ref.astype(np.float64)
    l2 = np.sqrt((d ** 2).sum(-1))
    return dict(maxabs=float(np.abs(d).max()), rmse=float(np.sqrt((d ** 2).mean())), mean_l2=float(l2.mean()),
                frac=float((l2 < 1e-4).mean()))


def stat_int(s):
    """StatsPrint abbreviates large numbers ('14.0k'); returns (value, exact?)."""
    s = s.strip()
    if s.endswith("k"):
        return float(s[:-1]) * 1e3, False
    if s.endswith("M"):
        return float(s[:-1]) * 1e6, False
    return float(s), True
