import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
REF = g.load_ref_runner()
from pbrt_v1_amd import scenes
blob = scenes.icosphere((200,120,250),90,1)
vi = '"single" "float stepsize" [80]'
cfgs = {
 "direct_point_only": dict(integrator="directlighting", world_kwargs=dict(volume=' ', point_light=True, area_light=False)),
 "direct_two_lights": dict(integrator="directlighting", world_kwargs=dict(volume=' ', point_light=True)),
 "direct_g03": dict(integrator="directlighting", world_kwargs=dict(volume='"float g" [.3]')),
 "direct_glass": dict(integrator="directlighting", world_kwargs=dict(volume=' ', glass_sphere_tris=blob)),
 "whitted_glass": dict(integrator="whitted", world_kwargs=dict(volume=' ', glass_sphere_tris=blob)),
 "whitted_mirror": dict(integrator="whitted", world_kwargs=dict(volume=' ', mirror_quad=True)),
 "whitted_glass_emission": dict(integrator="whitted", volume_integrator='"emission" "float stepsize" [80]', world_kwargs=dict(volume='"color Le" [.001 .001 .001]', glass_sphere_tris=blob)),
}
for name, kw in cfgs.items():
    kw = dict(kw); kw.setdefault("volume_integrator", vi)
    text = scenes.cornell_scene(xres=32, yres=32, keyed=True, count=True, **kw)
    rgb, alpha, cnt, ms = pkg.render_text(text)
    ref, ra, st = REF.run_reference(text, keyed=True)
    d = rgb - ref; l2 = np.sqrt((d**2).sum(-1))
    print(name, "maxabs %.3g frac %.4f" % (np.abs(d).max(), (l2 < 1e-4).mean()), "rays", cnt["closest_rays"], st["closest_rays"], cnt["any_rays"], st["any_rays"], flush=True)
