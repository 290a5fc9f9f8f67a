import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
from pbrt_v1_amd import scenes
import oracle
kw = dict(xres=1024, yres=1024, xsamples=4, ysamples=4, jitter=False, pixel_filter="box", soup_tris=1_000_000, keyed=True,
          integrator="directlighting", volume_integrator='"single" "float stepsize" [20]', world_kwargs=dict(volume='"float g" [0]'))
ps = pkg.ParsedScene(text=scenes.cornell_scene(**kw))
ds = pkg.DeviceScene(ps)
nodes, refs = ds.accel_arrays(); info = ds.accel_info(); bounds = np.array(list(info.bounds), np.float32)
for (row, col) in ((74, 555), (173, 322), (196, 495)):
    p = row * 1025 + col
    tile = p // 16
    ps.set_shard(tile % 65536, 65536, 16)
    res = {}
    for tag, env in (("pipe", dict(PBRT_HIP_PIPELINE="1")), ("mega", dict(PBRT_HIP_PIPELINE="0"))):
        os.environ.update(env)
        ds.set_counting(True); ds.reset_counters(); ds.clear_film() if res else None
        ds.render()
        res[tag] = (ds.film_accum().copy(), ds.counters())
    _, _, oacc, ocnt = oracle.render(ps, nodes, refs, bounds, info=info)
    for tag in ("pipe", "mega"):
        a, c = res[tag]
        print((row, col), tag, "pixel", a[:, row, col], "vs oracle", oacc[:, row, col], "film==oracle", np.array_equal(a, oacc), "maxdiff", float(np.abs(a - oacc).max()),
              {k: (c[k], ocnt[k]) for k in ("camera_rays", "closest_rays", "any_rays", "nodes_visited", "tri_tests")})
