"""pbrt-v1 Scene::Render hot path on MI355X -- Python harness over the C ABI.

This package is *plumbing*: ctypes bindings of ``lib/libpbrt_hip.so`` (include/pbrt_hip.h,
the HIP kernels + C ABI) and ``lib/libpbrt_host.so`` (the C++ host mirror of pbrt-v1's scene
API), used by tests/, bench.py and __graft_entry__.py.  There is no CPU fallback anywhere:
if the HIP library is missing or no GPU is visible, calls raise.

The directory name ``pbrt-v1_amd`` is not an importable identifier; load it with
``__graft_entry__.load_package()`` (importlib, module name ``pbrt_v1_amd``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HIP_LIB = os.path.join(LIB_DIR, "libpbrt_hip.so")
if os.environ.get("PBRT_HIP_TUNE") and os.environ.get("PBRT_HIP_LIB_PATH"):      # A/B measurements: a variant built by tools/build_variant.py
    HIP_LIB = os.environ["PBRT_HIP_LIB_PATH"]
HOST_LIB = os.path.join(LIB_DIR, "libpbrt_host.so")
CSRC = os.path.join(_HERE, "csrc")

# -fno-slp-vectorize: with the SLP vectorizer on, hipcc (ROCm 7.2) packs the y / z components of the integrators' 3-vectors into
# v_pk_*_f32 pairs, and the megakernel of the volume workload then computes (x, 0, 0) for 3 of 16.8 M camera samples of the C5 frame
# (deterministic, every flavour; tests/test_gpu_configs.py::test_pipeline_workloads_full_size_properties found it, the pipeline
# and the oracle agree on the right value; any added printf hides it).  Without it the results are right AND the kernels are faster
# (C3 kernel 16.8 -> 14.6 ms, C2 59.0 -> 57.4 ms: the packing costs registers).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-Wno-unused-value"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-msse2", "-mfpmath=sse", "-fvisibility=hidden", "-fvisibility-inlines-hidden"]


HIP_UNITS = ("rt_kernels.hip", "rt_scene.hip", "rt_film.hip", "rt_trace.hip", "rt_mega_w.hip", "rt_mega_d.hip", "rt_mega_dw.hip", "rt_mega_p.hip", "rt_pipe_w.hip", "rt_pipe_d.hip",
             "rt_pipe_p.hip", "rt_pipe_v.hip", "rt_march.hip", "kd_build.cpp", "grid_build.cpp", "leaf_layout.cpp")


def build(force: bool = False, verbose: bool = False, defines=(), jobs: int | None = None) -> None:
    """Compile both shared libraries in-tree (hipcc cross-compiles gfx950 without a GPU).  The HIP library is ten translation
    units (the megakernel and the pipeline's shade kernel per integrator, the trace kernel, the C ABI, the two host builders)
    compiled in parallel and linked; `defines` (-D knobs; tools/build_variant.py builds variants side by side) force a rebuild into the same place."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hip_dir = os.path.join(CSRC, "hip")
    hip_dep = [os.path.join(hip_dir, f) for f in os.listdir(hip_dir)] + [os.path.join(_HERE, "..", "include", "pbrt_hip.h")]
    headers = [d for d in hip_dep if d.endswith((".h", ".inc"))]
    host_src = [os.path.join(CSRC, "host", "scene_api.cpp")]
    host_dep = [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host"))] + \
               [os.path.join(_HERE, "..", "include", f) for f in ("pbrt_hip.h", "pbrt_hip_desc.h", "pbrt_hip_plugin.h")]

    def stale(out, deps):
        return force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)

    def closure(path, seen=None):
        """the files `path` includes with #include "..." (transitively): a unit is rebuilt only when one of ITS headers changed"""
        import re
        seen = set() if seen is None else seen
        if path in seen or not os.path.exists(path):
            return seen
        seen.add(path)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), re.M):
            closure(os.path.normpath(os.path.join(os.path.dirname(path), inc)), seen)
        return seen

    stamp = os.path.join(obj_dir, "defines.txt")
    # the stamp holds the compiler flags too: an object built with other flags (e.g. before -fno-slp-vectorize, a correctness fix) must
    # not survive an incremental build just because it is newer than its sources
    dtext = " ".join(HIPCC_FLAGS) + " | " + " ".join(HOST_FLAGS) + " | " + " ".join(defines)
    if (open(stamp).read() if os.path.exists(stamp) else "") != dtext:
        force = True
    todo = []
    for u in HIP_UNITS:
        obj = os.path.join(obj_dir, u.rsplit(".", 1)[0] + ".o")
        if stale(obj, sorted(closure(os.path.join(hip_dir, u)))):
            todo.append(["hipcc"] + [f for f in HIPCC_FLAGS if f != "-shared"] + list(defines) + ["-c", os.path.join(hip_dir, u), "-o", obj])
    if todo:
        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            if not cmd[-3].endswith(".hip"):
                subprocess.check_call(cmd)
                return
            # keep the compiler's per-kernel resource report (VGPRs, spills, occupancy) next to the object: tests/test_abi.py checks
            # that the timed kernels stay inside their occupancy step (gfx950: <= 168 VGPRs for 3 waves per SIMD, <= 64 for 8)
            r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
            keep = [ln for ln in r.stderr.splitlines() if "remark:" in ln and any(k in ln for k in ("Function Name", " VGPRs:", "VGPRs Spill", "Occupancy", "ScratchSize", "LDS Size"))]
            open(cmd[-1][:-2] + ".resources.txt", "w").write("\n".join(keep) + "\n")
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(run, todo))
        open(stamp, "w").write(dtext)
    objs = [os.path.join(obj_dir, u.rsplit(".", 1)[0] + ".o") for u in HIP_UNITS]
    if stale(HIP_LIB, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", HIP_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if stale(HOST_LIB, host_dep):
        cmd = ["g++"] + HOST_FLAGS + host_src + ["-Wl,--version-script=" + os.path.join(CSRC, "host", "exports.map"), "-ldl", "-o", HOST_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)


# --------------------------------------------------------------------------- C ABI structs
class RtCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("camera_rays", "closest_rays", "any_rays", "nodes_visited", "leaf_refs",
                                          "tri_tests", "bad_samples", "stack_overflows")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RtAccelInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaf_refs", C.c_uint32), ("max_depth", C.c_uint32), ("n_tris", C.c_uint32),
                ("bounds", C.c_float * 6), ("build_seconds", C.c_double), ("kind", C.c_int32),
                ("grid_nvoxels", C.c_int32 * 3), ("grid_width", C.c_float * 3), ("grid_inv_width", C.c_float * 3)]


class RtLeafLayoutInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_slots", C.c_uint64), ("n_entries", C.c_uint64), ("stride", C.c_uint32), ("pad", C.c_uint32)]


class RtPrebuiltAccel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_nodes", C.c_uint32), ("n_leaf_refs", C.c_uint32), ("max_depth", C.c_uint32),
                ("nodes", C.c_void_p), ("leaf_refs", C.c_void_p), ("bounds", C.c_float * 6),
                ("grid_nvoxels", C.c_int32 * 3), ("grid_width", C.c_float * 3), ("grid_inv_width", C.c_float * 3)]


class RtRenderStats(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("render_ms", C.c_float), ("trace_ms", C.c_float), ("gather_ms", C.c_float),
                ("pipeline", C.c_int32), ("iterations", C.c_int32), ("timed_iterations", C.c_int32), ("slots", C.c_uint32), ("shade_ms", C.c_float), ("bands", C.c_int32), ("march_ms", C.c_float),
                ("weighted_points", C.c_uint64), ("weighted_ms", C.c_float * 5)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n == "weighted_ms" else getattr(self, n)) for n, _ in self._fields_}


RAY_DTYPE = np.dtype([("o", np.float32, 3), ("d", np.float32, 3), ("mint", np.float32), ("maxt", np.float32)])
HIT_DTYPE = np.dtype([("prim", np.int32), ("t", np.float32), ("b1", np.float32), ("b2", np.float32)])

_hip = None
_host = None


def code_id(path: str | None = None) -> str:
    """Identity of the DEVICE code in libpbrt_hip.so: sha256 (16 hex digits) of its .hip_fatbin section (the gfx950 code objects of every
    kernel).  tools/publish_profile.py stores it with the PMC summaries under profiles/, bench.py compares it with the library it has
    loaded and refuses to print counters that were collected on other kernels."""
    import hashlib, struct
    data = open(path or HIP_LIB, "rb").read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise RuntimeError("not a 64-bit ELF file: " + (path or HIP_LIB))
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sh(i):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size
    _, stroff, _ = sh(shstrndx)
    h = hashlib.sha256()
    found = False
    for i in range(shnum):
        name, off, size = sh(i)
        end = data.index(b"\0", stroff + name)
        if data[stroff + name:end] == b".hip_fatbin":
            h.update(data[off:off + size]); found = True
    if not found:
        raise RuntimeError("no .hip_fatbin section in " + (path or HIP_LIB))
    return h.hexdigest()[:16]


def hip_lib():
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError("libpbrt_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(HIP_LIB)
        L.rt_last_error.restype = C.c_char_p
        for name in ("rt_scene_create", "rt_scene_destroy", "rt_scene_set_stream", "rt_scene_accel_info",
                     "rt_scene_accel_copy", "rt_camera_rays", "rt_trace_closest", "rt_trace_any", "rt_film_bind",
                     "rt_film_clear", "rt_film_read", "rt_film_resolve", "rt_render", "rt_sync", "rt_counters",
                     "rt_counters_reset", "rt_last_render_ms", "rt_last_render_stats", "rt_samples_read", "rt_device_count", "rt_set_counting",
                     "rt_kdtree_build", "rt_kdtree_info", "rt_kdtree_copy", "rt_kdtree_destroy",
                     "rt_accel_build", "rt_accel_info", "rt_accel_copy", "rt_accel_destroy", "rt_scene_create_prebuilt", "rt_film_resolve_device",
                     "rt_film_resolve_device_rgba", "rt_film_pack_parts", "rt_accel_leaf_layout"):
            getattr(L, name).restype = C.c_int
        L.rt_scene_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_scene_create_prebuilt.argtypes = [C.c_void_p, C.c_int, C.POINTER(RtPrebuiltAccel), C.POINTER(C.c_void_p)]
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_scene_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_scene_accel_info.argtypes = [C.c_void_p, C.POINTER(RtAccelInfo)]
        L.rt_scene_accel_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.rt_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.rt_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.rt_film_bind.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.rt_film_clear.argtypes = [C.c_void_p]
        L.rt_film_read.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_film_resolve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rt_render.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_film_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        L.rt_film_resolve_device_rgba.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        L.rt_film_pack_parts.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.rt_sync.argtypes = [C.c_void_p]
        L.rt_counters.argtypes = [C.c_void_p, C.POINTER(RtCounters)]
        L.rt_counters_reset.argtypes = [C.c_void_p]
        L.rt_last_render_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.rt_last_render_stats.argtypes = [C.c_void_p, C.POINTER(RtRenderStats)]
        L.rt_samples_read.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.rt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.rt_set_counting.argtypes = [C.c_void_p, C.c_int]
        L.rt_kdtree_build.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.rt_kdtree_info.argtypes = [C.c_void_p, C.POINTER(RtAccelInfo)]
        L.rt_kdtree_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_kdtree_destroy.argtypes = [C.c_void_p]
        L.rt_accel_build.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.rt_accel_info.argtypes = [C.c_void_p, C.POINTER(RtAccelInfo)]
        L.rt_accel_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_accel_destroy.argtypes = [C.c_void_p]
        L.rt_accel_leaf_layout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RtLeafLayoutInfo)]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("libpbrt_host.so is not built (run __graft_entry__.build())")
        L = C.CDLL(HOST_LIB)
        L.pbrt_host_parse_file.restype = C.c_void_p
        L.pbrt_host_parse_file.argtypes = [C.c_char_p, C.c_int]
        L.pbrt_host_parse_string.restype = C.c_void_p
        L.pbrt_host_parse_string.argtypes = [C.c_char_p, C.c_int]
        L.pbrt_host_free.argtypes = [C.c_void_p]
        L.pbrt_host_frame_count.argtypes = [C.c_void_p]
        L.pbrt_host_frame_valid.argtypes = [C.c_void_p, C.c_int]
        L.pbrt_host_scene_desc.restype = C.c_void_p
        L.pbrt_host_scene_desc.argtypes = [C.c_void_p, C.c_int]
        L.pbrt_host_render_desc.restype = C.c_void_p
        L.pbrt_host_render_desc.argtypes = [C.c_void_p, C.c_int]
        L.pbrt_host_premultiply.argtypes = [C.c_void_p, C.c_int]
        L.pbrt_host_filename.restype = C.c_char_p
        L.pbrt_host_filename.argtypes = [C.c_void_p, C.c_int]
        L.pbrt_host_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.pbrt_host_set_seed.argtypes = [C.c_void_p, C.c_uint]
        L.pbrt_host_film_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int * 8)]
        L.pbrt_host_scene_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint * 4)]
        L.pbrt_host_camera.restype = C.POINTER(C.c_float)
        L.pbrt_host_camera.argtypes = [C.c_void_p]
        L.pbrt_host_tri_verts.restype = C.POINTER(C.c_float)
        L.pbrt_host_tri_verts.argtypes = [C.c_void_p]
        L.pbrt_host_write_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6
        L.pbrt_host_read_exr_info.argtypes = [C.c_char_p, C.POINTER(C.c_int * 6)]
        L.pbrt_host_read_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        L.pbrt_host_format_f32.restype = C.c_longlong
        L.pbrt_host_format_f32.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_longlong]
        L.pbrt_host_format_iota.restype = C.c_longlong
        L.pbrt_host_format_iota.argtypes = [C.c_longlong, C.c_longlong, C.c_int, C.c_void_p, C.c_longlong]
        L.pbrt_host_accel_params.restype = C.c_void_p
        L.pbrt_host_accel_params.argtypes = [C.c_void_p]
        _host = L
    return _host


def write_exr(path, rgb, alpha, total_res=None, offset=(0, 0)):
    """WriteRGBAImage (core/exrio.cpp:75-96): RGBA half EXR with data/display windows (own minimal writer)."""
    rgb = np.ascontiguousarray(rgb, np.float32); alpha = np.ascontiguousarray(alpha, np.float32)
    h, w = alpha.shape
    tw, th = total_res if total_res else (w, h)
    if host_lib().pbrt_host_write_exr(path.encode(), rgb.ctypes.data, alpha.ctypes.data, w, h, tw, th, offset[0], offset[1]) != 0:
        raise IOError("cannot write " + path)


def read_exr(path):
    """ReadImage for the files write_exr / WriteRGBAImage produce: uncompressed half RGBA scanline EXR.  LIMIT: compressed files are
    refused (a message on stderr names the compression) -- in particular the PIZ files a stock pbrt-v1 linked against OpenEXR writes
    (RgbaOutputFile's default); nothing in this image can produce one to validate a decoder against."""
    info = (C.c_int * 6)()
    if host_lib().pbrt_host_read_exr_info(path.encode(), C.byref(info)) != 0:
        raise IOError("cannot read " + path)
    w, h = info[0], info[1]
    rgb = np.zeros((h, w, 3), np.float32); alpha = np.zeros((h, w), np.float32)
    host_lib().pbrt_host_read_exr(path.encode(), rgb.ctypes.data, alpha.ctypes.data)
    return rgb, alpha, dict(total_res=(info[2], info[3]), offset=(info[4], info[5]))


TRI_SHADING_DTYPE = np.dtype([("flags", np.uint32), ("xform", np.uint32), ("uv", np.float32, 6), ("n", np.float32, 9), ("s", np.float32, 9)])


def shading_records(parsed):
    """(tri_shading[n_tris] or None, RtTriShading records, xforms[n][32]) of a ParsedScene: the per-vertex uv / N / S data."""
    H = host_lib()
    H.pbrt_host_shading.restype = C.c_int
    H.pbrt_host_shading.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    idx, rec, xf, nrec, nxf = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint(), C.c_uint()
    H.pbrt_host_shading(parsed.scene_desc, C.byref(idx), C.byref(rec), C.byref(xf), C.byref(nrec), C.byref(nxf))
    if not idx.value:
        return None, np.zeros(0, TRI_SHADING_DTYPE), np.zeros((0, 32), np.float32)
    i = np.ctypeslib.as_array(C.cast(idx, C.POINTER(C.c_int32)), shape=(parsed.n_tris,)).copy()
    r = np.frombuffer(C.string_at(rec, nrec.value * TRI_SHADING_DTYPE.itemsize), TRI_SHADING_DTYPE).copy()
    x = np.ctypeslib.as_array(C.cast(xf, C.POINTER(C.c_float)), shape=(nxf.value, 32)).copy() if nxf.value else np.zeros((0, 32), np.float32)
    return i, r, x


def format_f32(values, per_line: int = 9) -> str:
    """Scene-text numbers ("%.9g": round-trips float32) formatted by the host library: 1 s for a 1 M-triangle mesh."""
    v = np.ascontiguousarray(values, np.float32).ravel()
    buf = C.create_string_buffer(int(v.size) * 17 + 64)
    n = host_lib().pbrt_host_format_f32(v.ctypes.data, v.size, per_line, buf, len(buf))
    if n < 0:
        raise RuntimeError("format_f32: buffer too small")
    return buf.raw[:n].decode()


def format_iota(first: int, count: int, per_line: int = 3) -> str:
    buf = C.create_string_buffer(int(count) * 12 + 64)
    n = host_lib().pbrt_host_format_iota(first, count, per_line, buf, len(buf))
    if n < 0:
        raise RuntimeError("format_iota: buffer too small")
    return buf.raw[:n].decode()


def assemble_exr(paths, out) -> float:
    """tools/exrassemble.cpp: merge crop-window EXRs into the display-window image; returns the covered fraction.  LIMIT: the inputs
    must be uncompressed half RGBA scanline files (this library's own output); PIZ tiles written by a stock pbrt are refused, see read_exr."""
    H = host_lib()
    H.pbrt_host_assemble_exr.restype = C.c_float
    H.pbrt_host_assemble_exr.argtypes = [C.c_char_p, C.c_char_p]
    r = H.pbrt_host_assemble_exr("\n".join(paths).encode(), out.encode())
    if r < 0:
        raise IOError("cannot assemble " + out)
    return float(r)


def publish_accel(ds: "DeviceScene", path: str) -> None:
    """Rank 0 of a node: write the accelerator of `ds` where the other ranks can map it (e.g. under /dev/shm): one file, header
    (RtAccelInfo bytes) + nodes + leaf refs, renamed into place when complete."""
    info = ds.accel_info()
    need = 256 + 8 * int(info.n_nodes) + 4 * int(info.n_leaf_refs)
    st = os.statvfs(os.path.dirname(path) or ".")
    if st.f_bavail * st.f_frsize < need + (64 << 20):
        raise OSError("publish_accel: %s has %d MB free, the accelerator needs %d MB" % (os.path.dirname(path), st.f_bavail * st.f_frsize >> 20, need >> 20))
    nodes, refs = ds.accel_arrays()
    tmp = path + ".tmp%d" % os.getpid()
    try:
        with open(tmp, "wb") as f:
            f.write(bytes(info).ljust(256, b"\0"))
            np.ascontiguousarray(nodes, np.uint32).tofile(f); np.ascontiguousarray(refs, np.uint32).tofile(f)     # (no second copy in memory: 6 GB at 10 M triangles)
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise


def attach_accel(path: str):
    """The other ranks: (nodes, leaf_refs, RtAccelInfo) mapped read-only from a file written by publish_accel."""
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    info = RtAccelInfo.from_buffer_copy(bytes(raw[:C.sizeof(RtAccelInfo)]))
    nodes = np.frombuffer(raw, np.uint32, count=2 * info.n_nodes, offset=256).reshape(-1, 2)
    refs = np.frombuffer(raw, np.uint32, count=info.n_leaf_refs, offset=256 + 8 * info.n_nodes)
    return nodes, refs, info


class RtError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise RtError("rt error %d: %s" % (rc, hip_lib().rt_last_error().decode()))


def device_count() -> int:
    n = C.c_int(0)
    rc = hip_lib().rt_device_count(C.byref(n))
    return int(n.value) if rc == 0 else 0


def build_kdtree(tri_verts: np.ndarray, accel_params_ptr=None):
    """Host-only kd-tree build (rt_kdtree_build).  Returns (nodes[n,2] u32, leaf_refs u32, bounds[6], info)."""
    tv = np.ascontiguousarray(tri_verts, np.float32).reshape(-1, 9)
    t = C.c_void_p()
    _chk(hip_lib().rt_accel_build(tv.ctypes.data, len(tv), accel_params_ptr, C.byref(t)))
    info = RtAccelInfo()
    _chk(hip_lib().rt_accel_info(t, C.byref(info)))
    nodes = np.zeros((info.n_nodes, 2), np.uint32)
    refs = np.zeros(max(info.n_leaf_refs, 1), np.uint32)
    _chk(hip_lib().rt_accel_copy(t, nodes.ctypes.data, refs.ctypes.data))
    hip_lib().rt_accel_destroy(t)
    return nodes, refs[:info.n_leaf_refs], np.array(list(info.bounds), np.float32), info


def leaf_layout(tri_verts: np.ndarray, accel_params_ptr=None, runs: bool = False, copies: bool = False):
    """Host-only: the kd-tree of `tri_verts` and its leaves as rt_scene_create lays them out for the flat traversal (rt_accel_leaf_layout).
    Returns (nodes[n,2], leaf_refs, tnodes[n,2], slot_prim, entries, stride)."""
    tv = np.ascontiguousarray(tri_verts, np.float32).reshape(-1, 9)
    t = C.c_void_p()
    _chk(hip_lib().rt_accel_build(tv.ctypes.data, len(tv), accel_params_ptr, C.byref(t)))
    try:
        info = RtAccelInfo()
        _chk(hip_lib().rt_accel_info(t, C.byref(info)))
        nodes = np.zeros((info.n_nodes, 2), np.uint32)
        refs = np.zeros(max(info.n_leaf_refs, 1), np.uint32)
        _chk(hip_lib().rt_accel_copy(t, nodes.ctypes.data, refs.ctypes.data))
        li = RtLeafLayoutInfo()
        _chk(hip_lib().rt_accel_leaf_layout(t, int(runs), int(copies), None, None, None, C.byref(li)))
        tn = np.zeros((li.n_nodes, 2), np.uint32); sp = np.zeros(max(li.n_slots, 1), np.uint32); en = np.zeros(max(li.n_entries, 1), np.uint32)
        _chk(hip_lib().rt_accel_leaf_layout(t, int(runs), int(copies), tn.ctypes.data, sp.ctypes.data, en.ctypes.data, C.byref(li)))
    finally:
        hip_lib().rt_accel_destroy(t)
    return nodes, refs[:info.n_leaf_refs], tn, sp[:li.n_slots], en[:li.n_entries], int(li.stride)


def fit_tile(extent: int, size: int) -> int:
    """The tile size in [3/4, 5/4] of `size` whose whole tiles overshoot `extent` least (ties: the one nearest `size`)."""
    best = None
    for t in range(max(1, size * 3 // 4), max(2, size * 5 // 4) + 1):
        pad = -(-extent // t) * t - extent
        key = (pad, abs(t - size))
        if best is None or key < best[0]:
            best = (key, t)
    return best[1]


class ParsedScene:
    """A .pbrt description run through the host API mirror (scene_api.cpp): flat scene + frame descriptors."""

    def __init__(self, text: str | None = None, path: str | None = None, quiet: bool = True, frame: int = 0):
        H = host_lib()
        if path is not None:
            self._h = H.pbrt_host_parse_file(path.encode(), int(quiet))
        else:
            from . import scenes as _scenes
            text = _scenes.for_product(text)      # texts written for runs of the compiled reference wrap the sampler / accelerator in oracle-side plugins
            self._h = H.pbrt_host_parse_string(text.encode(), int(quiet))
        self.warnings = H.pbrt_host_warnings()
        self.errors = H.pbrt_host_errors()
        self.n_frames = H.pbrt_host_frame_count(self._h)
        if self.n_frames == 0:
            raise RtError("scene description has no WorldEnd (errors: %d)" % self.errors)
        self.frame = frame
        self.valid = bool(H.pbrt_host_frame_valid(self._h, frame))
        self.scene_desc = H.pbrt_host_scene_desc(self._h, frame)
        self.render_desc = H.pbrt_host_render_desc(self._h, frame)
        self.premultiply = bool(H.pbrt_host_premultiply(self._h, frame))
        dims = (C.c_int * 8)()
        H.pbrt_host_film_dims(self.render_desc, C.byref(dims))
        self.width, self.height = dims[0], dims[1]
        self.sample_extent = (dims[2], dims[3], dims[4], dims[5])
        self.spp = dims[6]
        self.integrator = dims[7]
        cnt = (C.c_uint * 4)()
        H.pbrt_host_scene_counts(self.scene_desc, C.byref(cnt))
        self.n_tris, self.n_materials, self.n_lights, self.n_light_tris = (int(c) for c in cnt)

    @property
    def n_camera_samples(self) -> int:
        x0, x1, y0, y1 = self.sample_extent
        return (x1 - x0) * (y1 - y0) * self.spp

    def set_shard(self, index: int, count: int, tile_pixels=64, fit: bool = False):
        """tile_pixels: an int = tiles of that many consecutive pixels of the sample extent (scanline order); a pair (w, h) = 2-D tiles of
        w x h pixels (RtRenderDesc.tile_pixels = -(w | h << 16)).  fit: adjust w and h (within 3/4 .. 5/4 of the request) to the sizes that
        pad the sample extent least -- border tiles are whole tiles whose pixels outside the extent are fetched and dropped, a lane idling for
        about a ray's time per dropped work item (64 x 64 tiles pad a 1025 x 1025 extent by 12.7 %, 61 x 61 by 2.4 %).  Every rank of a job
        computes the same sizes.  Returns the tile size used."""
        if isinstance(tile_pixels, (tuple, list)):
            w, h = int(tile_pixels[0]), int(tile_pixels[1])
            if fit:
                x0, x1, y0, y1 = self.sample_extent
                w, h = fit_tile(x1 - x0, w), fit_tile(y1 - y0, h)
            used = (w, h)
            tile_pixels = -(w | (h << 16))
        else:
            used = int(tile_pixels)
        host_lib().pbrt_host_set_shard(self.render_desc, index, count, int(tile_pixels))
        return used

    def set_seed(self, seed: int):
        host_lib().pbrt_host_set_seed(self.render_desc, seed)

    def camera_matrices(self):
        p = host_lib().pbrt_host_camera(self.scene_desc)
        a = np.ctypeslib.as_array(p, shape=(32,)).copy()
        return a[:16].reshape(4, 4), a[16:].reshape(4, 4)

    def accel_params_ptr(self):
        return host_lib().pbrt_host_accel_params(self.scene_desc)

    def accel_params(self) -> dict:
        """RtAccelParams of this frame (include/pbrt_hip.h)."""
        a = np.ctypeslib.as_array(C.cast(self.accel_params_ptr(), C.POINTER(C.c_int32)), shape=(7,))
        return {"kind": int(a[0]), "isect_cost": int(a[1]), "trav_cost": int(a[2]), "max_prims": int(a[3]), "max_depth": int(a[4]),
                "empty_bonus": float(a[5:6].view(np.float32)[0]), "build_threads": int(a[6])}

    def render_view(self) -> dict:
        """The leading scalar fields of this frame's RtRenderDesc (include/pbrt_hip.h): integrator .. seed."""
        a = np.ctypeslib.as_array(C.cast(self.render_desc, C.POINTER(C.c_int32)), shape=(11,))
        return {"integrator": int(a[0]), "max_depth": int(a[1]), "strategy": int(a[2]), "volume_integrator": int(a[3]),
                "step_size": float(a[4:5].view(np.float32)[0]), "sampler": int(a[5]), "x_samples": int(a[6]), "y_samples": int(a[7]),
                "jitter": int(a[8]), "pixel_samples": int(a[9]), "seed": int(a[10:11].view(np.uint32)[0])}

    def kdtree(self):
        """The kd-tree rt_scene_create would build for this scene, built on the host only."""
        return build_kdtree(self.tri_verts(), self.accel_params_ptr())

    def tri_verts(self):
        if self.n_tris == 0:
            return np.zeros((0, 3, 3), np.float32)
        p = host_lib().pbrt_host_tri_verts(self.scene_desc)
        return np.ctypeslib.as_array(p, shape=(self.n_tris, 3, 3)).copy()

    def serialize(self) -> bytes:
        """Canonical byte image of this frame's descriptors (include/pbrt_hip_desc.h rt_desc_serialize)."""
        H = host_lib()
        H.pbrt_host_serialize.restype = C.c_longlong
        H.pbrt_host_serialize.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
        n = H.pbrt_host_serialize(self._h, self.frame, None, 0)
        buf = C.create_string_buffer(int(n))
        H.pbrt_host_serialize(self._h, self.frame, buf, n)
        return buf.raw

    def close(self):
        if self._h:
            host_lib().pbrt_host_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceScene:
    """rt_scene_create: kd-tree build + upload.  Holds the film unless one is bound externally."""

    def __init__(self, parsed: ParsedScene, device: int = -1, prebuilt=None):
        """prebuilt = (nodes[n][2] uint32, leaf_refs uint32, RtAccelInfo) of an accelerator built elsewhere (another rank's
        DeviceScene.accel_arrays() / accel_info(), e.g. through publish_accel / attach_accel): rt_scene_create_prebuilt."""
        self.parsed = parsed
        self._s = C.c_void_p()
        if prebuilt is None:
            _chk(hip_lib().rt_scene_create(parsed.scene_desc, device, C.byref(self._s)))
        else:
            nodes, refs, info = prebuilt
            nodes = np.ascontiguousarray(nodes, np.uint32); refs = np.ascontiguousarray(refs if len(refs) else np.zeros(1, np.uint32), np.uint32)
            pa = RtPrebuiltAccel(kind=info.kind, n_nodes=info.n_nodes, n_leaf_refs=info.n_leaf_refs, max_depth=info.max_depth,
                                 nodes=nodes.ctypes.data, leaf_refs=refs.ctypes.data)
            for i in range(6): pa.bounds[i] = info.bounds[i]
            for i in range(3):
                pa.grid_nvoxels[i] = info.grid_nvoxels[i]; pa.grid_width[i] = info.grid_width[i]; pa.grid_inv_width[i] = info.grid_inv_width[i]
            _chk(hip_lib().rt_scene_create_prebuilt(parsed.scene_desc, device, C.byref(pa), C.byref(self._s)))
        self._film_bound = False

    def accel_info(self) -> RtAccelInfo:
        info = RtAccelInfo()
        _chk(hip_lib().rt_scene_accel_info(self._s, C.byref(info)))
        return info

    def accel_arrays(self):
        info = self.accel_info()
        nodes = np.zeros((info.n_nodes, 2), np.uint32)
        refs = np.zeros(max(info.n_leaf_refs, 1), np.uint32)
        _chk(hip_lib().rt_scene_accel_copy(self._s, nodes.ctypes.data, refs.ctypes.data))
        return nodes, refs[:info.n_leaf_refs]

    def set_stream(self, stream_ptr: int):
        _chk(hip_lib().rt_scene_set_stream(self._s, C.c_void_p(stream_ptr)))

    def bind_film(self, device_ptr: int | None = None):
        _chk(hip_lib().rt_film_bind(self._s, C.c_void_p(device_ptr) if device_ptr else None,
                                     self.parsed.width, self.parsed.height))
        self._film_bound = True

    def clear_film(self):
        _chk(hip_lib().rt_film_clear(self._s))

    def render(self, sync: bool = True):
        if not self._film_bound:
            self.bind_film()
        _chk(hip_lib().rt_render(self._s, self.parsed.render_desc))
        if sync:
            self.sync()

    def sync(self):
        _chk(hip_lib().rt_sync(self._s))

    def resolve_device(self, accum_ptr: int, n: int, rgb_ptr: int, alpha_ptr: int, premultiply: bool | None = None):
        """rt_film_resolve_device: WriteImage's normalisation of a 5-plane accumulator (planes of n floats) in device memory."""
        pm = self.parsed.premultiply if premultiply is None else premultiply
        _chk(hip_lib().rt_film_resolve_device(self._s, C.c_void_p(accum_ptr), n, int(pm), C.c_void_p(rgb_ptr), C.c_void_p(alpha_ptr)))

    def resolve_device_rgba(self, accum_ptr: int, n: int, rgba_ptr: int, premultiply: bool | None = None):
        """rt_film_resolve_device_rgba: the same, interleaved RGBA [n][4] (the payload of one all-gather)."""
        pm = self.parsed.premultiply if premultiply is None else premultiply
        _chk(hip_lib().rt_film_resolve_device_rgba(self._s, C.c_void_p(accum_ptr), n, int(pm), C.c_void_p(rgba_ptr)))

    def pack_parts(self, accum_ptr: int, world: int, rows: int, parts_ptr: int):
        """rt_film_pack_parts: this rank's 5-plane film as `world` parts of `rows` rows, part r = [5][rows][W] (the send buffer of one reduce-scatter)."""
        _chk(hip_lib().rt_film_pack_parts(self._s, C.c_void_p(accum_ptr), self.parsed.width, self.parsed.height, world, rows, C.c_void_p(parts_ptr)))

    def film_accum(self) -> np.ndarray:
        out = np.zeros((5, self.parsed.height, self.parsed.width), np.float32)
        _chk(hip_lib().rt_film_read(self._s, out.ctypes.data))
        return out

    def film(self, premultiply: bool | None = None, out=None):
        """rt_film_resolve: the film as ImageFilm::WriteImage hands it to the image writer.  `out` = (rgb[H,W,3], alpha[H,W])
        float32 C-contiguous arrays to fill (e.g. views of page-locked memory, reused across frames); allocated if omitted."""
        h, w = self.parsed.height, self.parsed.width
        if out is None:
            rgb = np.empty((h, w, 3), np.float32); alpha = np.empty((h, w), np.float32)
        else:
            rgb, alpha = out
            if rgb.shape != (h, w, 3) or alpha.shape != (h, w) or rgb.dtype != np.float32 or alpha.dtype != np.float32 \
                    or not rgb.flags.c_contiguous or not alpha.flags.c_contiguous:
                raise ValueError("film(out=...): need C-contiguous float32 arrays of shape (H,W,3) and (H,W)")
        pm = self.parsed.premultiply if premultiply is None else premultiply
        _chk(hip_lib().rt_film_resolve(self._s, int(pm), rgb.ctypes.data, alpha.ctypes.data))
        return rgb, alpha

    def counters(self) -> dict:
        c = RtCounters()
        _chk(hip_lib().rt_counters(self._s, C.byref(c)))
        return c.as_dict()

    def set_counting(self, enabled: bool):
        _chk(hip_lib().rt_set_counting(self._s, int(enabled)))

    def reset_counters(self):
        _chk(hip_lib().rt_counters_reset(self._s))

    def last_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        _chk(hip_lib().rt_last_render_ms(self._s, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def samples(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """The per-camera-sample records of the last render: [n][8] = L.rgb, alpha, imageX, imageY, -, - in the sampler's order."""
        n = self.parsed.n_camera_samples - first if count is None else count
        out = np.zeros((n, 8), np.float32)
        _chk(hip_lib().rt_samples_read(self._s, first, n, out.ctypes.data))
        return out

    def last_stats(self) -> dict:
        st = RtRenderStats()
        _chk(hip_lib().rt_last_render_stats(self._s, C.byref(st)))
        return st.as_dict()

    def camera_rays(self, first: int, count: int) -> np.ndarray:
        rays = np.zeros(count, RAY_DTYPE)
        _chk(hip_lib().rt_camera_rays(self._s, self.parsed.render_desc, first, count, rays.ctypes.data))
        return rays

    def trace_closest(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        _chk(hip_lib().rt_trace_closest(self._s, rays.ctypes.data, len(rays), hits.ctypes.data))
        return hits

    def trace_any(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        occ = np.zeros(len(rays), np.uint8)
        _chk(hip_lib().rt_trace_any(self._s, rays.ctypes.data, len(rays), occ.ctypes.data))
        return occ

    def close(self):
        if self._s:
            hip_lib().rt_scene_destroy(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_text(scene_text: str, device: int = -1):
    """Convenience: parse -> create -> render -> (rgb, alpha, counters)."""
    ps = ParsedScene(text=scene_text)
    ds = DeviceScene(ps, device)
    ds.render()
    rgb, alpha = ds.film()
    cnt = ds.counters()
    ms = ds.last_ms()[0]
    ds.close()
    return rgb, alpha, cnt, ms
