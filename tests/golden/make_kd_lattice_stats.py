"""Generates tests/golden/chain/kd_lattice30k.npz: the UNMODIFIED reference (oracle/_ref/pbrt_ref) builds its KdTreeAccel over 30 000 triangles whose
vertices sit on a lattice -- nearly every bounding-box edge ties with many others, so the order std::sort leaves tied edges in (kdtree.cpp:246) decides
which primitives share leaves -- and prints its own StatsPrint table (core/util.cpp).  The fixture = the triangles (data) + the printed values.
kd_build.cpp defines the tie order as (t, START < END, primitive number): tests/test_oracle_golden.py compares its tree's counts with
this table, so a drift between the defined order and what the reference's sort does on tie-heavy input is visible.  Authoring container only.
usage: python tests/golden/make_kd_lattice_stats.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from pbrt_v1_amd import scenes
runner = entry.load_ref_runner()
rng = np.random.default_rng(3)
n = 30000
centre = np.floor(rng.uniform(0, 556, (n, 1, 3)) / 16) * 16
tv = (centre + np.floor(rng.uniform(-8, 8, (n, 3, 3)) / 4) * 4).astype(np.float32)
text = scenes.options_block(xres=4, yres=4, integrator="whitted") + 'WorldBegin\nLightSource "point" "point from" [278 450 279.5] "color I" [400000 400000 400000]\n' + \
    'Material "matte" "color Kd" [.5 .5 .5]\n' + scenes.soup_shape_text(tv) + "WorldEnd\n"
t0 = time.time()
_, _, st = runner.run_reference(text, keyed=False, timeout=3600)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "chain", "kd_lattice30k.npz"), stats=json.dumps(st["stats"]), tri_verts=tv.reshape(n, 9))
print("kd_lattice30k.npz %.1f s" % (time.time() - t0), st["stats"], flush=True)
