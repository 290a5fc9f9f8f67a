#!/usr/bin/env python3
"""Build a variant of libpbrt_hip.so next to the product library for A/B measurements in ONE gpurun call.

    python tools/build_variant.py NAME [-DKNOB=V ...] [--mllvm OPT ...] [--units rt_trace.hip,rt_mega_p.hip]

compiles the listed translation units (default: all) with the extra defines into pbrt-v1_amd/lib/obj_NAME/, takes the others
from the product build (pbrt-v1_amd/lib/obj/), and links pbrt-v1_amd/lib/libpbrt_hip_NAME.so.  A process picks it with
PBRT_HIP_LIB_PATH=<that file> (read by pbrt-v1_amd/__init__.py only when PBRT_HIP_TUNE is set, like every other knob)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry


def main():
    pkg = entry.load_package()
    name = sys.argv[1]
    defines = [a for a in sys.argv[2:] if a.startswith("-D")]
    for i, a in enumerate(sys.argv):                              # --mllvm OPT: a code generation option for the device compiler (scheduling experiments)
        if a == "--mllvm":
            defines += ["-mllvm", sys.argv[i + 1]]
    arch = None
    for i, a in enumerate(sys.argv):                              # --arch gfx950:xnack- : another target id for the device code (code generation experiments)
        if a == "--arch":
            arch = sys.argv[i + 1]
    units = list(pkg.HIP_UNITS)
    for i, a in enumerate(sys.argv):
        if a == "--units":
            units = sys.argv[i + 1].split(",")
    pkg.build()                                                  # the product objects the variant borrows
    obj_dir = os.path.join(pkg.LIB_DIR, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    hip_dir = os.path.join(pkg.CSRC, "hip")

    def run(u):
        obj = os.path.join(obj_dir, u.rsplit(".", 1)[0] + ".o")
        flags = [f for f in pkg.HIPCC_FLAGS if f != "-shared"]
        if arch:
            flags = ["--offload-arch=" + arch if f.startswith("--offload-arch=") else f for f in flags]
        cmd = ["hipcc"] + flags + defines + ["-c", os.path.join(hip_dir, u), "-o", obj]
        if not u.endswith(".hip"):
            subprocess.check_call(cmd)
            return obj
        r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)      # the per-kernel resource report, as the product build keeps it
        keep = [ln.split("remark:")[1].split("[-R")[0].rstrip() for ln in r.stderr.splitlines() if "remark:" in ln and any(k in ln for k in ("Function Name", " VGPRs:", "VGPRs Spill", "Occupancy", "ScratchSize", "LDS Size"))]
        open(obj[:-2] + ".resources.txt", "w").write("\n".join(keep) + "\n")
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
        return obj
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 4)) as ex:
        built = dict(zip(units, ex.map(run, units)))
    objs = [built.get(u, os.path.join(pkg.LIB_DIR, "obj", u.rsplit(".", 1)[0] + ".o")) for u in pkg.HIP_UNITS]
    out = os.path.join(pkg.LIB_DIR, "libpbrt_hip_%s.so" % name)
    subprocess.check_call(["hipcc", "--offload-arch=" + (arch or "gfx950"), "-shared", "-fPIC"] + objs + ["-o", out])
    print(out)


if __name__ == "__main__":
    main()
