import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
from pbrt_v1_amd import scenes
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ntri = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
kw = dict(xres=res, yres=res, xsamples=4, ysamples=4, jitter=False, pixel_filter="box", soup_tris=ntri, keyed=True,
          integrator="directlighting", volume_integrator='"single" "float stepsize" [20]', world_kwargs=dict(volume='"float g" [0]'))
ps = pkg.ParsedScene(text=scenes.cornell_scene(**kw))
ds = pkg.DeviceScene(ps)
out = {}
for tag, env, counting in (("pipe_count", dict(PBRT_HIP_PIPELINE="1"), True), ("pipe", dict(PBRT_HIP_PIPELINE="1"), False), ("pipe_1M", dict(PBRT_HIP_PIPELINE="1", PBRT_HIP_PIPE_SLOTS="1048576"), False),
                           ("mega_count", dict(PBRT_HIP_PIPELINE="0"), True), ("mega_occ1", dict(PBRT_HIP_PIPELINE="0", PBRT_HIP_HIGH_OCC="1"), False),
                           ("mega_occ0", dict(PBRT_HIP_PIPELINE="0", PBRT_HIP_HIGH_OCC="0"), False), ("mega_count2", dict(PBRT_HIP_PIPELINE="0"), True)):
    for k in ("PBRT_HIP_PIPELINE", "PBRT_HIP_HIGH_OCC", "PBRT_HIP_PIPE_SLOTS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ds.set_counting(counting); ds.reset_counters()
    if out: ds.clear_film()
    ds.render()
    out[tag] = (ds.film_accum().copy(), ds.counters())
ref = out["pipe_count"][0]
for tag, (a, c) in out.items():
    d = np.abs(a - ref)
    bad = np.argwhere(d.max(0) > 0)
    print(tag, "equal", np.array_equal(a, ref), "npix_diff", len(bad), "maxdiff", float(d.max()), "first", bad[:5].tolist(), {k: c[k] for k in ("camera_rays", "closest_rays", "any_rays", "bad_samples")})
