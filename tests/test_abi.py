"""The C-ABI shared library: it loads, it exports every symbol include/pbrt_hip.h declares, and without a GPU it
fails loudly instead of falling back to anything.  CPU only (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pbrt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    syms = declared_symbols()
    for s in ("rt_scene_create", "rt_scene_destroy", "rt_render", "rt_trace_closest", "rt_trace_any", "rt_camera_rays",
              "rt_film_bind", "rt_film_read", "rt_film_resolve", "rt_counters", "rt_kdtree_build", "rt_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol(pkg):
    lib = C.CDLL(pkg.HIP_LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_no_torch_or_cxx_types_in_signatures():
    text = open(os.path.join(ROOT, "include", "pbrt_hip.h")).read()
    assert "std::" not in text and "torch" not in text.lower().replace("torch tensor", "")
    assert 'extern "C"' in text


def test_scene_create_fails_loudly_without_gpu(pkg, scenes):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible")
    ps = pkg.ParsedScene(text=scenes.cornell_scene(xres=8, yres=8))
    with pytest.raises(pkg.RtError) as e:
        pkg.DeviceScene(ps)
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under pbrt-v1_amd/ or bench.py's timed path may reach it."""
    pkg_dir = os.path.join(ROOT, "pbrt-v1_amd")
    offenders = []
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"#include\s+\"[^\"]*oracle/|import oracle|from oracle|libpbrt_oracle", src):
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders


def test_kdtree_build_host_only(pkg, scenes):
    tris = scenes.lcg_soup(300)
    nodes, refs, bounds, info = pkg.build_kdtree(tris.reshape(-1, 9))
    assert info.n_tris == 300 and info.n_nodes == len(nodes) and info.n_nodes % 2 == 1   # full binary tree
    leaf = (nodes[:, 0] & 3) == 3
    assert leaf.sum() == (~leaf).sum() + 1
    assert np.all(bounds[:3] <= tris.reshape(-1, 3).min(0)) and np.all(bounds[3:] >= tris.reshape(-1, 3).max(0))
    # every triangle is referenced by at least one leaf
    np_leaf = nodes[leaf, 0] >> 2
    single = nodes[leaf][np_leaf == 1][:, 1]
    assert set(single.tolist()) | set(refs.tolist()) == set(range(300))
    # interior nodes point forward; reference depth bound Round2Int(8 + 1.3*Log2Int(N)) (kdtree.cpp:159-161)
    idx = np.nonzero(~leaf)[0]
    assert np.all(nodes[idx, 1] > idx + 1) and info.max_depth == 18


def test_timed_kernels_stay_inside_their_occupancy_step(pkg):
    """gfx950 allocates VGPRs in steps: <= 64 -> 8 waves per SIMD, <= 128 -> 4, <= 168 -> 3, more -> 2 (MI355X_MICROARCH.md).  The
    C2 headline kernel lives 2 registers below the 3-wave step (a 3-register change costs 30 % of the frame rate: measured in round 2),
    the trace kernel of the queue pipeline runs 6 waves per SIMD (its 24 KB of LDS stack planes per workgroup set that), i.e. <= 80 VGPRs.  The compiler's own resource report is kept at build time
    (pbrt-v1_amd/lib/obj/*.resources.txt)."""
    import re
    pkg.build()                                   # incremental: a no-op when the objects are current, and then the reports are the ones of THESE objects
    def vgprs(unit, mangled_fragment):
        rep, obj = os.path.join(pkg.LIB_DIR, "obj", unit + ".resources.txt"), os.path.join(pkg.LIB_DIR, "obj", unit + ".o")
        # the report is written by the very compiler invocation that writes the object (pkg.build), after it: never older than the object
        assert os.path.exists(rep) and os.path.exists(obj) and os.path.getmtime(rep) >= os.path.getmtime(obj) - 1.0, "stale resource report for " + unit
        text = open(rep).read()
        blocks = re.split(r"remark: Function Name: ", text)[1:]
        for b in blocks:
            if mangled_fragment in b.splitlines()[0]:
                return int(re.search(r" VGPRs: (\d+)", b).group(1)), int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        raise AssertionError("kernel %s not in %s report" % (mangled_fragment, unit))
    v, spill = vgprs("rt_mega_p", "render_kernelILb0ELi2ELi0ELb0ELi1ELb0EE")      # path, kd-tree, no volume, natural allocation, no EXT
    assert v <= 168 and spill == 0, (v, spill)
    v, spill = vgprs("rt_mega_d", "render_kernelILb0ELi1ELi0ELb0ELi3ELb0EE")      # directlighting, kd-tree: held to the 3-wave step (170 on its own)
    assert v <= 168 and spill <= 8, (v, spill)
    v, spill = vgprs("rt_mega_w", "render_kernelILb0ELi0ELi0ELb0ELi1ELb0EE")      # whitted
    assert v <= 168 and spill == 0, (v, spill)
    v, spill = vgprs("rt_trace", "pipe_trace_kernelILb0ELi0ELb0EE")
    assert v <= 80 and spill == 0, (v, spill)
    v, spill = vgprs("rt_mega_p", "render_kernelILb0ELi2ELi0ELb0ELi4ELb0EE")      # the 4-waves flavour is capped at 128
    assert v <= 128, v
    v, spill = vgprs("rt_pipe_v", "pipe_vertex_kernelILb0ELb0EE")                 # the by-vertex shade pass of the path pipeline: 3 waves
    assert v <= 168 and spill == 0, (v, spill)
    v, spill = vgprs("rt_march", "pipe_march_kernelILb0ELi0ELb0EE")               # the march kernel (round 4): 4 waves; a handful of spilled registers outside its traversal loop
    assert v <= 128 and spill <= 24, (v, spill)
    v, spill = vgprs("rt_pipe_d", "pipe_shade_kernelILb0ELi1ELb1ELb0EE")          # the volume shade pass without the march (round 3: 195 on its own, held to 168)
    assert v <= 168, (v, spill)
    # the film gathers keep their pixel windows in registers: a dynamic window index (or an `if` the compiler turns into a store through a
    # pointer phi) puts them in scratch, 112 B per lane -- seen while writing film_march_kernel
    for frag, cap in (("film_slot_kernelILi2ELi2ELi4EE", 168), ("film_slot_kernelILi2ELi2ELi12EE", 256), ("film_slot_kernelILi1ELi1ELi4EE", 168),
                      ("film_march_kernelILi2EE", 128)):
        v, spill = vgprs("rt_film", frag)
        rep = open(os.path.join(pkg.LIB_DIR, "obj", "rt_film.resources.txt")).read()
        blk = [b for b in re.split(r"remark: Function Name: ", rep)[1:] if frag in b.splitlines()[0]][0]
        assert v <= cap and spill == 0 and int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1)) == 0, (frag, v, spill)
