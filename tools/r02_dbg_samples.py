import re, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
from pbrt_v1_amd import scenes
text = scenes.cornell_scene(xres=40, yres=30, integrator="directlighting", xsamples=2, ysamples=1, jitter=True, soup_tris=600, pixel_filter="gaussian",
                            world_kwargs=dict(light_nsamples=3, point_light=True))
text = re.sub(r'Sampler "(\w+)"', r'Sampler "keyed" "string inner" ["\1"] "integer seed" [3]', text)
ps = pkg.ParsedScene(text=text); ds = pkg.DeviceScene(ps); ds.render(); smp = ds.samples()
def pcg(v):
    v = np.uint32(v); s = np.uint32(v * np.uint32(747796405) + np.uint32(2891336453))
    w = np.uint32(((s >> np.uint32((s >> np.uint32(28)) + np.uint32(4))) ^ s) * np.uint32(277803737))
    return np.uint32((w >> np.uint32(22)) ^ w)
np.seterr(over="ignore")
x0, x1, y0, y1 = ps.sample_extent; w = x1 - x0
for n in (8, 10, 11, 20):
    pixel, s = n // 2, n % 2
    px, py = x0 + pixel % w, y0 + pixel // w
    base = pcg(np.uint32(pixel * 2) + np.uint32(3) * np.uint32(0x9E3779B9))
    jx = np.float32(pcg(np.uint32(2 * s) + base) & np.uint32(0xffffff)) / np.float32(1 << 24)
    jy = np.float32(pcg(np.uint32(2 * s + 1) + base) & np.uint32(0xffffff)) / np.float32(1 << 24)
    ix = np.float32(np.float32(np.float32(s % 2) + jx) * np.float32(0.5)) + np.float32(px)
    iy = np.float32(np.float32(np.float32(s // 2) + jy) * np.float32(1.0)) + np.float32(py)
    print(n, "device", float(smp[n, 4]).hex(), float(smp[n, 5]).hex(), "ieee", float(ix).hex(), float(iy).hex(), "jx", float(jx).hex())
