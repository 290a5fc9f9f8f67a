import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
from pbrt_v1_amd import scenes
import test_gpu_parity as T
k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
text = T._random_scene(scenes, np.random.default_rng(1000 + k))
print(text[:1500])
ps = pkg.ParsedScene(text=text)
res = {}
for tag, env, counting in (("mega_count", dict(PBRT_HIP_PIPELINE="0"), True), ("pipe_ray_count", dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_PIPE_VERTEX="0"), True),
                           ("default_count", dict(), True), ("default", dict(), False)):
    for kk in ("PBRT_HIP_PIPELINE", "PBRT_HIP_PIPE_VERTEX"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    ds = pkg.DeviceScene(ps)
    ds.set_counting(counting); ds.render()
    res[tag] = (ds.film_accum().copy(), ds.counters(), ds.last_stats())
    ds.close()
ref = res["mega_count"][0]
for tag, (acc, cnt, st) in res.items():
    d = np.abs(acc - ref)
    print(tag, "equal", np.array_equal(acc, ref), "maxdiff", float(d.max()), "ndiff", int((d > 0).sum()), st["pipeline"], st["iterations"], cnt)
