"""Print film metrics of the device render of golden fixtures (GPU box). usage: python tools/fixture_metrics.py name [name...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from conftest import load_golden, film_metrics
for name in sys.argv[1:]:
    gd = load_golden(name)
    ps = pkg.ParsedScene(text=gd["scene"])
    ds = pkg.DeviceScene(ps); ds.render(); rgb, alpha = ds.film(); cnt = ds.counters(); ds.close()
    m = film_metrics(rgb, gd["rgb"])
    print(name, {k: (round(v, 7) if isinstance(v, float) else v) for k, v in m.items()}, "closest", cnt["closest_rays"], gd["stats"]["closest_rays"],
          "any", cnt["any_rays"], gd["stats"]["any_rays"], flush=True)
    if os.environ.get("SHOW_BAD"):
        d = np.sqrt(((rgb - gd["rgb"]) ** 2).sum(axis=2))
        ys, xs = np.nonzero(d > 1e-4)
        print("bad pixels", len(ys), "bbox x", xs.min() if len(xs) else None, xs.max() if len(xs) else None, "y", ys.min() if len(ys) else None, ys.max() if len(ys) else None)
        for y, x in list(zip(ys, xs))[:12]:
            print(" ", x, y, rgb[y, x], gd["rgb"][y, x])
