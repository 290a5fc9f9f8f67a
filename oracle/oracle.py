"""ctypes binding of the CPU restatement (oracle/pbrt_oracle.cpp).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libpbrt_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(LIB)
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_trace.restype = C.c_int
        L.oracle_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_set_mailbox.restype = None
        L.oracle_set_mailbox.argtypes = [C.c_int]
        L.oracle_resolve.restype = None
        L.oracle_resolve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def set_mailbox(mode: int):
    """-1: per-primitive mailboxes like the reference; 0: none (default, like the HIP kernels); 4: four-entry window."""
    lib().oracle_set_mailbox(int(mode))


COUNTER_NAMES = ("camera_rays", "closest_rays", "any_rays", "nodes_visited", "leaf_refs", "tri_tests", "bad_samples", "tri_tests_closest")


def render(parsed, nodes=None, leaf_refs=None, bounds=None, premultiply=None, info=None):
    """Render `parsed` (a pbrt_v1_amd.ParsedScene: flat RtSceneDesc/RtRenderDesc) on one CPU thread.
    nodes/leaf_refs/bounds: a flattened kd-tree (e.g. from pbrt_v1_amd.build_kdtree); None = brute force.
    Returns (rgb, alpha, accum, counters)."""
    L = lib()
    w, h = parsed.width, parsed.height
    accum = np.zeros((5, h, w), np.float32)
    cnt = np.zeros(8, np.uint64)
    n_nodes = 0 if nodes is None else len(nodes)
    nd = None if nodes is None else np.ascontiguousarray(nodes, np.uint32)
    lr = None if leaf_refs is None else np.ascontiguousarray(leaf_refs if len(leaf_refs) else np.zeros(1, np.uint32), np.uint32)
    bd = None if bounds is None else np.ascontiguousarray(bounds, np.float32)
    rc = L.oracle_render(parsed.scene_desc, parsed.render_desc, None if nd is None else nd.ctypes.data, n_nodes,
                         None if lr is None else lr.ctypes.data, None if bd is None else bd.ctypes.data,
                         accum.ctypes.data, cnt.ctypes.data, C.byref(info) if info is not None else None)
    if rc != 0:
        raise RuntimeError("oracle_render failed: %d" % rc)
    rgb = np.zeros((h, w, 3), np.float32)
    alpha = np.zeros((h, w), np.float32)
    pm = parsed.premultiply if premultiply is None else premultiply
    L.oracle_resolve(accum.ctypes.data, w, h, int(pm), rgb.ctypes.data, alpha.ctypes.data)
    return rgb, alpha, accum, dict(zip(COUNTER_NAMES, (int(c) for c in cnt)))


def resolve(accum, premultiply=True):
    accum = np.ascontiguousarray(accum, np.float32)
    _, h, w = accum.shape
    rgb = np.zeros((h, w, 3), np.float32)
    alpha = np.zeros((h, w), np.float32)
    lib().oracle_resolve(accum.ctypes.data, w, h, int(premultiply), rgb.ctypes.data, alpha.ctypes.data)
    return rgb, alpha


def trace(parsed, rays, any_hit=False, nodes=None, leaf_refs=None, bounds=None, info=None):
    L = lib()
    rays = np.ascontiguousarray(rays)
    n = len(rays)
    hits = np.zeros(n, np.dtype([("prim", np.int32), ("t", np.float32), ("b1", np.float32), ("b2", np.float32)]))
    occ = np.zeros(n, np.uint8)
    cnt = np.zeros(8, np.uint64)
    n_nodes = 0 if nodes is None else len(nodes)
    nd = None if nodes is None else np.ascontiguousarray(nodes, np.uint32)
    lr = None if leaf_refs is None else np.ascontiguousarray(leaf_refs if len(leaf_refs) else np.zeros(1, np.uint32), np.uint32)
    bd = None if bounds is None else np.ascontiguousarray(bounds, np.float32)
    rc = L.oracle_trace(parsed.scene_desc, None if nd is None else nd.ctypes.data, n_nodes, None if lr is None else lr.ctypes.data,
                        None if bd is None else bd.ctypes.data, rays.ctypes.data, n, int(any_hit), hits.ctypes.data, occ.ctypes.data,
                        cnt.ctypes.data, C.byref(info) if info is not None else None)
    if rc != 0:
        raise RuntimeError("oracle_trace failed")
    return (occ if any_hit else hits), dict(zip(COUNTER_NAMES, (int(c) for c in cnt)))
