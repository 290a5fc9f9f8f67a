"""Generates tests/golden/chain/kd_soup60k.npz and kd_soup1m.npz: the UNMODIFIED reference (oracle/_ref/pbrt_ref, built from
/root/reference by oracle/ref/Makefile) builds its KdTreeAccel over the Cornell box + N-triangle LCG soup and prints its own
statistics (core/util.cpp StatsPrint: nodes, leaves, primitives per leaf).  Runs only in the authoring container (needs
/root/reference to have been compiled into oracle/_ref); the fixture is the table of printed values, nothing else.
usage: python tests/golden/make_kd_stats.py [60000] [1000000]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from pbrt_v1_amd import scenes
runner = entry.load_ref_runner()
for n in [int(a) for a in sys.argv[1:]] or [60000]:
    text = scenes.cornell_scene(xres=4, yres=4, integrator="whitted", soup_tris=n, world_kwargs=dict(point_light=True, area_light=False))
    t0 = time.time()
    _, _, st = runner.run_reference(text, keyed=False, timeout=3600)
    name = "kd_soup%s.npz" % ("%dk" % (n // 1000) if n < 1_000_000 else "%dm" % (n // 1_000_000))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "chain", name), stats=json.dumps(st["stats"]), soup_tris=n)
    print(name, "%.1f s" % (time.time() - t0), st["stats"], flush=True)
