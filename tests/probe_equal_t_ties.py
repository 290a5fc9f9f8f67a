#!/usr/bin/env python3
"""GPU box: where does the reference trace one more ray on the tuned tree than on the default tree (c3_tuned parity leg: 16920456 vs 16920455, films equal)?
Camera rays of the crop through rt_trace_closest on both trees and through the CPU oracle on the tuned tree: hits that differ between trees are equal-t ties
(the kept primitive depends on leaf order, trianglemesh.cpp:245); device and oracle must agree on the SAME tree."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))      # the checker: this diagnostic lives under tests/ because only tests may use oracle/
os.environ.setdefault("PBRT_HIP_TUNE", "1")
import numpy as np
import __graft_entry__ as entry
import bench
pkg = entry.load_package()
import oracle
out = {}
for name in ("c3", "c3_tuned"):
    _, _, crop = bench.workload(name)
    ps = pkg.ParsedScene(text=bench.oracle_side_text(name, crop, keyed=True))
    ds = pkg.DeviceScene(ps)
    n = ps.n_camera_samples
    hits = []
    rays_all = []
    for lo in range(0, n, 1 << 21):
        r = ds.camera_rays(lo, min(1 << 21, n - lo))
        rays_all.append(r); hits.append(ds.trace_closest(r))
    out[name] = (np.concatenate(rays_all), np.concatenate(hits), ps, ds)
    print(name, n, "camera rays traced", flush=True)
r0, h0, _, _ = out["c3"]; r1, h1, ps1, ds1 = out["c3_tuned"]
assert np.array_equal(r0, r1)
d = np.nonzero((h0["prim"] != h1["prim"]) | (h0["t"] != h1["t"]))[0]
print("camera rays whose closest hit differs between the two trees:", len(d))
for i in d[:20]:
    print("  ray", i, "default", h0[i], "tuned", h1[i])
# the oracle on the tuned tree, for the differing rays and a sample of the others
nodes, refs = ds1.accel_arrays(); info = ds1.accel_info(); bounds = np.array(list(info.bounds), np.float32)
idx = np.unique(np.concatenate([d, np.arange(0, len(r1), 97)]))
oh, _ = oracle.trace(ps1, r1[idx], nodes=nodes, leaf_refs=refs, bounds=bounds, info=info)
bad = np.nonzero((oh["prim"] != h1["prim"][idx]) | (oh["t"] != h1["t"][idx]) | (oh["b1"] != h1["b1"][idx]) | (oh["b2"] != h1["b2"][idx]))[0]
print("device vs oracle on the tuned tree, %d rays: %d differ" % (len(idx), len(bad)))
for k in bad[:20]:
    print("  ray", idx[k], "device", h1[idx[k]], "oracle", oh[k])
