import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
from pbrt_v1_amd import scenes
order = sys.argv[1].split(",")
kw = dict(xres=1024, yres=1024, xsamples=4, ysamples=4, jitter=False, pixel_filter="box", soup_tris=1_000_000, keyed=True,
          integrator="directlighting", volume_integrator='"single" "float stepsize" [20]', world_kwargs=dict(volume='"float g" [0]'))
ps = pkg.ParsedScene(text=scenes.cornell_scene(**kw))
ds = pkg.DeviceScene(ps)
row, col = 196, 495
tile = (row * 1025 + col) // 16
ps.set_shard(tile % 65536, 65536, 16)
for tag in order:
    os.environ["PBRT_HIP_PIPELINE"] = "1" if tag == "pipe" else "0"
    ds.set_counting(tag.endswith("c")); ds.render()
    print(order, tag, ds.samples(0, 256)[51][:4])
