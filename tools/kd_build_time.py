#!/usr/bin/env python3
"""Host-side kd build time (kd_build.cpp) on the N-triangle soup of the benchmark scenes, per thread count.  No GPU needed.
    python tools/kd_build_time.py N_TRIS [THREADS ...]        (THREADS: 0 = auto, -1 = the sorting form, one thread)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry


class P(C.Structure):
    _fields_ = [("kind", C.c_int32), ("isect_cost", C.c_int32), ("trav_cost", C.c_int32), ("max_prims", C.c_int32), ("max_depth", C.c_int32),
                ("empty_bonus", C.c_float), ("build_threads", C.c_int32)]


def main():
    pkg = entry.load_package()
    from pbrt_v1_amd import scenes
    n = int(sys.argv[1])
    threads = [int(a) for a in sys.argv[2:]] or [0]
    ps = pkg.ParsedScene(text=scenes.cornell_scene(xres=4, yres=4, integrator="whitted", soup_tris=n, world_kwargs=dict(point_light=True, area_light=False)))
    tv = np.ascontiguousarray(ps.tri_verts(), np.float32).reshape(-1, 9)
    first = None
    for t in threads:
        p = P(0, 80, 1, 1, -1, 0.5, t)
        t0 = time.time()
        nodes, refs, bounds, info = pkg.build_kdtree(tv, C.addressof(p))
        same = "" if first is None else (" same arrays as the first: %s" % (np.array_equal(first[0], nodes) and np.array_equal(first[1], refs)))
        if first is None:
            first = (nodes, refs)
        print("%d triangles, build_threads %3d: build %.3f s (call %.3f s), %d nodes, %d leaf refs, depth %d%s" % (len(tv), t, info.build_seconds, time.time() - t0, len(nodes), len(refs), info.max_depth, same), flush=True)
        del nodes, refs


if __name__ == "__main__":
    main()
