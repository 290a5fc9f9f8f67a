"""Runner for the compiled, unmodified reference (oracle/_ref, built from /root/reference by oracle/ref/Makefile).
TEST INFRASTRUCTURE like the rest of oracle/: used only by tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never imported by the product package pbrt-v1_amd."""
from __future__ import annotations

import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_ref_film(path: str):
    """Read a float film dumped by oracle/ref/ref_driver.cpp (WriteRGBAImage tap)."""
    b = open(path, "rb").read()
    if b[:8] != b"PBRTFILM":
        raise ValueError("not a PBRTFILM dump: " + path)
    h = np.frombuffer(b[8:32], np.int32)
    n = int(h[0]) * int(h[1])
    rgb = np.frombuffer(b[32:32 + 12 * n], np.float32).reshape(h[1], h[0], 3).copy()
    alpha = np.frombuffer(b[32 + 12 * n:32 + 16 * n], np.float32).reshape(h[1], h[0]).copy()
    return rgb, alpha, h


REF_DIR = os.path.join(_HERE, "_ref")


def run_reference(scene_text: str, keyed: bool = True, workdir: str | None = None, timeout: float = 3600, env: dict | None = None):
    """Run the compiled reference (oracle/_ref) on a scene; returns (rgb, alpha, stats dict).
    TEST/BENCH infrastructure only -- never on the product path."""
    import json
    import tempfile
    exe = os.path.join(REF_DIR, "pbrt_ref_keyed" if keyed else "pbrt_ref")
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    d = workdir or tempfile.mkdtemp(prefix="pbrtref_")
    sp = os.path.join(d, "scene.pbrt")
    fp = os.path.join(d, "film.bin")
    with open(sp, "w") as f:
        f.write(scene_text if isinstance(scene_text, str) else "")
    env = dict(os.environ, PBRT_SEARCHPATH=os.path.join(REF_DIR, "bin"), **(env or {}))
    # a tuple ("builtin", kind, res) runs the hand-written API calls of ref_driver.cpp (BuiltinCornell) instead of a scene file
    args = [sp] if isinstance(scene_text, str) else ["--builtin", scene_text[1], str(scene_text[2])] + (["--keyed-sampler"] if keyed else [])
    r = subprocess.run([exe, "--out", fp] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=d)
    if r.returncode != 0:
        raise RuntimeError("reference run failed: %s\n%s" % (r.stdout[-2000:], r.stderr[-2000:]))
    lines = r.stdout.strip().splitlines()
    stats = json.loads(lines[-1])
    # the reference's own StatsPrint block (core/util.cpp:228-262), e.g. "Interior kd-tree nodes made   1625"
    table = {}
    if "Statistics:" in lines:
        for ln in lines[lines.index("Statistics:") + 1:-1]:
            if ln.startswith("    ") and len(ln.split()) >= 2:
                key = " ".join(ln.split()[:-1]) if not ln.rstrip().endswith(")") else " ".join(ln.split()[:-2])
                val = ln.split()[-1] if not ln.rstrip().endswith(")") else ln.split()[-2]
                table[key] = val
    stats["stats"] = table
    stats["stderr_lines"] = r.stderr.count("\n")
    stats["radiance_warnings"] = sum(r.stderr.count(k) for k in ("Not-a-number radiance", "Negative luminance", "Infinite luminance"))
    rgb, alpha, _ = load_ref_film(fp)
    return rgb, alpha, stats


def as_hip_plugin_scene(scene_text: str) -> str:
    """The same scene rendered through the reference-side binding oracle/ref/hip_adapter.cpp: the surface integrator and the
    accelerator become the "hip" plugin (which names the reference plugin it stands for as "inner")."""
    import re
    text = re.sub(r'SurfaceIntegrator "(whitted|directlighting|path)"', r'SurfaceIntegrator "hip" "string inner" ["\1"]', scene_text)
    text = re.sub(r'Accelerator "countaccel" "string inner" \["(\w+)"\]', r'Accelerator "hip" "string inner" ["\1"]', text)
    text = re.sub(r'Accelerator "(kdtree|grid)"', r'Accelerator "hip" "string inner" ["\1"]', text)
    assert '"hip"' in text
    return text


def reference_descriptors(scene_text: str, keyed: bool = False) -> bytes:
    """rt_desc_serialize() of the descriptors the adapter builds from the reference's own objects (PBRT_HIP_DESC_DUMP)."""
    import tempfile
    d = tempfile.mkdtemp(prefix="pbrtdesc_")
    dump = os.path.join(d, "desc.bin")
    run_reference(as_hip_plugin_scene(scene_text), keyed=keyed, workdir=d, env={"PBRT_HIP_DESC_DUMP": dump})
    return open(dump, "rb").read()
