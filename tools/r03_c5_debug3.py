import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
from pbrt_v1_amd import scenes
kw = dict(xres=1024, yres=1024, xsamples=4, ysamples=4, jitter=False, pixel_filter="box", soup_tris=1_000_000, keyed=True,
          integrator="directlighting", volume_integrator='"single" "float stepsize" [20]', world_kwargs=dict(volume='"float g" [0]'))
ps = pkg.ParsedScene(text=scenes.cornell_scene(**kw))
ds = pkg.DeviceScene(ps)
for (row, col) in ((74, 555), (196, 495)):
    p = row * 1025 + col
    tile = p // 16
    ps.set_shard(tile % 65536, 65536, 16)
    res = {}
    for tag, env in (("pipe", dict(PBRT_HIP_PIPELINE="1")), ("mega", dict(PBRT_HIP_PIPELINE="0")), ("mega_t1", dict(PBRT_HIP_PIPELINE="0", PBRT_HIP_TRAV_MODE="1")),
                     ("mega_noexit", dict(PBRT_HIP_PIPELINE="0", PBRT_HIP_EXIT_THRESH="0"))):
        for k in ("PBRT_HIP_TRAV_MODE", "PBRT_HIP_EXIT_THRESH"): os.environ.pop(k, None)
        os.environ.update(env)
        ds.set_counting(True); ds.render()
        n = 2 * 16 * 16 if tile + 65536 < (1025 * 1025 + 15) // 16 else 16 * 16
        res[tag] = ds.samples(0, 16 * 16).copy()
    a = res["pipe"]
    for tag in ("mega", "mega_t1", "mega_noexit"):
        b = res[tag]
        bad = np.argwhere(np.abs(a - b).max(1) > 1e-6)[:, 0]
        print((row, col), tag, "differing samples", bad.tolist())
        for i in bad[:4]:
            print("   sample", i, "pipe", a[i], tag, b[i])
