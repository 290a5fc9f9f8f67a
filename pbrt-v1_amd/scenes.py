"""Synthetic scene generators in pbrt-v1 scene syntax.

The reference ships no scenes at all, so the benchmark/parity inputs are defined here:
the classic Cornell box as six two-triangle ``trianglemesh`` quads with one diffuse area
light, and an N-triangle "soup" from a 32-bit LCG.  The text produced here is consumed
both by the MI355X host front end (pbrt-v1_amd/csrc/host) and -- in the authoring
container -- by the compiled reference (oracle/_ref/pbrt_ref*), so that both render the
very same scene description.
"""
from __future__ import annotations

import numpy as np

CORNELL_QUADS = [
    # (name, Kd, 4 vertices)
    ("floor", (.73, .73, .73), [552.8, 0, 0, 0, 0, 0, 0, 0, 559.2, 549.6, 0, 559.2]),
    ("ceiling", (.73, .73, .73), [556, 548.8, 0, 556, 548.8, 559.2, 0, 548.8, 559.2, 0, 548.8, 0]),
    ("back", (.73, .73, .73), [549.6, 0, 559.2, 0, 0, 559.2, 0, 548.8, 559.2, 556, 548.8, 559.2]),
    ("right", (.12, .45, .15), [0, 0, 559.2, 0, 0, 0, 0, 548.8, 0, 0, 548.8, 559.2]),
    ("left", (.65, .05, .05), [552.8, 0, 0, 549.6, 0, 559.2, 556, 548.8, 559.2, 556, 548.8, 0]),
]
CORNELL_LIGHT = [343, 548.7, 227, 343, 548.7, 332, 213, 548.7, 332, 213, 548.7, 227]


def _fmt(vals):
    return " ".join(repr(float(v)) if isinstance(v, (float, np.floating)) else str(v) for v in vals)


_SOUP_CACHE: dict = {}
_SOUP_TEXT_CACHE: dict = {}


def lcg_soup(n_tris: int, seed: int = 12345) -> np.ndarray:
    """Cached front of _lcg_soup (bench.py renders several workloads over the same 1 M-triangle soup)."""
    key = (int(n_tris), int(seed))
    if key not in _SOUP_CACHE:
        if len(_SOUP_CACHE) > 4:
            _SOUP_CACHE.clear(); _SOUP_TEXT_CACHE.clear()
        a = _lcg_soup(n_tris, seed)
        a.setflags(write=False)
        _SOUP_CACHE[key] = a
    return _SOUP_CACHE[key]


def _lcg_soup(n_tris: int, seed: int = 12345) -> np.ndarray:
    """N triangles, centres uniform in [50,500]x[50,450]x[50,500], vertices centre +- U(-4,4)
    per axis, from the LCG s = s*1664525 + 1013904223 (mod 2^32), u = (s >> 8) / 2^24.
    Returns float32 array [n_tris, 3, 3].  Vectorised: the LCG is jumped with the closed
    form for affine maps so that 10M triangles generate in seconds."""
    n = n_tris * 12
    a, c, m = 1664525, 1013904223, 1 << 32
    # doubling scheme: (A_k, C_k) such that s_{i+k} = A_k s_i + C_k
    if n == 0:
        return np.zeros((0, 3, 3), np.float32)
    s = np.empty(n, dtype=np.uint32)              # uint32 array arithmetic wraps mod 2^32, which is the LCG's modulus
    s[0] = (seed * a + c) % m
    filled = 1
    A, C = a, c
    with np.errstate(over="ignore"):
        while filled < n:
            k = min(filled, n - filled)
            s[filled:filled + k] = s[:k] * np.uint32(A) + np.uint32(C)
            filled += k
            C = (A * C + C) % m
            A = (A * A) % m
    u = (s >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / (1 << 24))     # < 2^24: exact in float32
    u = u.reshape(n_tris, 12)
    lo = np.array([50, 50, 50], np.float32)
    ext = np.array([450, 400, 450], np.float32)
    centre = lo + u[:, 0:3] * ext
    off = (u[:, 3:12].reshape(n_tris, 3, 3) * np.float32(8) - np.float32(4))
    return (centre[:, None, :] + off).astype(np.float32)


MAX_MESH_TRIS = 1_000_000


def soup_shape_text(tris: np.ndarray) -> str:
    """One `trianglemesh` per MAX_MESH_TRIS triangles: the pbrt-v1 file format reads every number as a float
    (pbrtlex.l: atof into a float), so vertex indices above 2^24 = 16.7 M cannot be written exactly -- a single mesh holds at
    most 5.59 M independent triangles in the reference too."""
    from . import format_f32, format_iota
    out = []
    for lo in range(0, tris.shape[0], MAX_MESH_TRIS):
        part = tris[lo:lo + MAX_MESH_TRIS]
        n = part.shape[0]
        out.append('Shape "trianglemesh" "integer indices" [%s] "point P" [%s]\n' % (format_iota(0, 3 * n, 3), format_f32(part, 9)))
    return "".join(out)


def _soup_block(soup: np.ndarray, soup_materials: bool) -> str:
    """The soup's Attribute block(s); the text of a cached lcg_soup is cached too (120 MB for 1 M triangles)."""
    ckey = None
    for k, v in _SOUP_CACHE.items():
        if v is soup:
            ckey = (k, bool(soup_materials))
    if ckey is not None and ckey in _SOUP_TEXT_CACHE:
        return _SOUP_TEXT_CACHE[ckey]
    out = []
    if not soup_materials:
        out.append("AttributeBegin # soup\n")
        out.append('  Material "matte" "color Kd" [.5 .5 .5]\n')
        out.append("  " + soup_shape_text(soup))
        out.append("AttributeEnd\n")
    else:
        # C4 material mix: tri % 10 == 0 -> glass(1.5), == 1 -> mirror, else matte
        n = soup.shape[0]
        cls = np.arange(n) % 10
        for label, sel, mat in (("glass", cls == 0, 'Material "glass" "float index" [1.5]'),
                                ("mirror", cls == 1, 'Material "mirror"'),
                                ("matte", cls >= 2, 'Material "matte" "color Kd" [.6 .55 .5]')):
            if sel.any():
                out.append("AttributeBegin # soup %s\n  %s\n  " % (label, mat))
                out.append(soup_shape_text(soup[sel]))
                out.append("AttributeEnd\n")
    text = "".join(out)
    if ckey is not None:
        _SOUP_TEXT_CACHE[ckey] = text
    return text


def cornell_world(soup: np.ndarray | None = None, soup_materials: bool = False,
                  point_light: bool = False, area_light: bool = True,
                  light_L=(17, 12, 4), light_nsamples: int = 1,
                  glass_sphere_tris: np.ndarray | None = None,
                  mirror_quad: bool = False, volume: str | None = None, extra: str = "") -> str:
    out = ["WorldBegin\n"]
    if extra:
        out.append(extra)
    if point_light:
        out.append('LightSource "point" "point from" [278 450 279.5] "color I" [400000 400000 400000]\n')
    for name, kd, verts in CORNELL_QUADS:
        out.append("AttributeBegin # %s\n" % name)
        out.append('  Material "matte" "color Kd" [%s]\n' % _fmt(kd))
        out.append('  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [%s]\n' % _fmt(verts))
        out.append("AttributeEnd\n")
    if area_light:
        out.append("AttributeBegin # light\n")
        out.append('  AreaLightSource "area" "color L" [%s] "integer nsamples" [%d]\n' % (_fmt(light_L), light_nsamples))
        out.append('  Material "matte" "color Kd" [0 0 0]\n')
        out.append('  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [%s]\n' % _fmt(CORNELL_LIGHT))
        out.append("AttributeEnd\n")
    if mirror_quad:
        out.append("AttributeBegin # mirror panel\n")
        out.append('  Material "mirror" "color Kr" [.9 .9 .9]\n')
        out.append('  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" '
                   '[100 50 400 300 50 500 300 350 500 100 350 400]\n')
        out.append("AttributeEnd\n")
    if glass_sphere_tris is not None:
        out.append("AttributeBegin # glass blob\n")
        out.append('  Material "glass" "float index" [1.5]\n')
        out.append("  " + soup_shape_text(glass_sphere_tris))
        out.append("AttributeEnd\n")
    if soup is not None and len(soup):
        out.append(_soup_block(soup, soup_materials))
    if volume:
        out.append('Volume "homogeneous" "point p0" [0 0 0] "point p1" [556 549 559] '
                   '"color sigma_a" [.002 .002 .002] "color sigma_s" [.002 .002 .002] %s\n' % volume)
    out.append("WorldEnd\n")
    return "".join(out)


def icosphere(center, radius, subdiv=2) -> np.ndarray:
    """Small triangulated sphere (for glass/mirror recursion tests)."""
    t = (1.0 + 5 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
                  [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
         (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
         (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    tris = np.array([[v[a], v[b], v[c]] for a, b, c in f])
    for _ in range(subdiv):
        a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        tris = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1),
                               np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    tris = tris / np.linalg.norm(tris, axis=2, keepdims=True)
    return (np.asarray(center, np.float64) + radius * tris).astype(np.float32)


def smooth_mesh_text(radius=80.0, nu=10, nv=6, with_n=True, with_uv=True, with_s=False, squash=(1.0, 1.0, 1.0), mirror_uv=False) -> str:
    """A latitude / longitude sphere as one `trianglemesh` with per-vertex "N" / "uv" / "S" (object space, centred at the
    origin): the input of Triangle::GetShadingGeometry (shapes/trianglemesh.cpp:71-133).  `squash` scales the positions only, so
    that the shading normals differ from the geometric ones by more than the tessellation; `mirror_uv` flips u (a mapping with
    negative determinant: dpdu x dpdv then points the other way)."""
    P, N, UV, S = [], [], [], []
    for j in range(nv + 1):
        th = np.pi * j / nv
        for i in range(nu + 1):
            ph = 2 * np.pi * i / nu
            d = np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
            P.append(radius * d * np.array(squash)); N.append(d)
            UV.append(((1.0 - i / nu) if mirror_uv else i / nu, j / nv))
            S.append((-np.sin(ph), np.cos(ph), 0.0))
    idx = []
    for j in range(nv):
        for i in range(nu):
            a, b = j * (nu + 1) + i, j * (nu + 1) + i + 1
            c, d = a + nu + 1, b + nu + 1
            if j > 0: idx += [a, c, b]
            if j < nv - 1: idx += [b, c, d]
    out = 'Shape "trianglemesh" "integer indices" [%s] "point P" [%s]' % (" ".join(map(str, idx)), _fmt(np.array(P, np.float32).ravel()))
    if with_n: out += ' "normal N" [%s]' % _fmt(np.array(N, np.float32).ravel())
    if with_uv: out += ' "float uv" [%s]' % _fmt(np.array(UV, np.float32).ravel())
    if with_s: out += ' "vector S" [%s]' % _fmt(np.array(S, np.float32).ravel())
    return out + "\n"


def options_block(xres=512, yres=512, integrator="whitted", integrator_params="", maxdepth=5,
                  sampler="stratified", xsamples=1, ysamples=1, jitter=False, pixelsamples=None,
                  pixel_filter="box", filter_params="", accelerator="kdtree", accel_params="",
                  keyed=False, count=False, seed=0, crop=None, fov=39.3, lensradius=0.0,
                  focaldistance=1e30, volume_integrator=None, film_name="out.exr") -> str:
    """Options block.  ``keyed``/``count`` wrap the sampler/accelerator in the oracle-side helper
    plugins (oracle/ref/keyed_sampler.cpp, count_accel.cpp) for runs of the compiled reference; for_product() below
    turns such a text into what the MI355X host is given (its RNG is always keyed, its rays are always counted)."""
    out = ["LookAt 278 273 -800  278 273 0  0 1 0\n"]
    cam = 'Camera "perspective" "float fov" [%s]' % repr(float(fov))
    if lensradius > 0:
        cam += ' "float lensradius" [%s] "float focaldistance" [%s]' % (repr(float(lensradius)), repr(float(focaldistance)))
    out.append(cam + "\n")
    film = 'Film "image" "integer xresolution" [%d] "integer yresolution" [%d] "string filename" ["%s"]' % (xres, yres, film_name)
    if crop is not None:
        film += ' "float cropwindow" [%s]' % _fmt([float(c) for c in crop])
    out.append(film + "\n")
    if sampler == "random":
        sp = '"integer xsamples" [%d] "integer ysamples" [%d]' % (xsamples, ysamples)
    elif sampler == "stratified":
        sp = '"integer xsamples" [%d] "integer ysamples" [%d] "bool jitter" ["%s"]' % (xsamples, ysamples, "true" if jitter else "false")
    else:
        sp = '"integer pixelsamples" [%d]' % (pixelsamples if pixelsamples is not None else xsamples * ysamples)
    if keyed:
        out.append('Sampler "keyed" "string inner" ["%s"] "integer seed" [%d] %s\n' % (sampler, seed, sp))
    else:
        out.append('Sampler "%s" %s\n' % (sampler, sp))
    out.append('PixelFilter "%s" %s\n' % (pixel_filter, filter_params))
    ip = integrator_params
    if integrator in ("whitted", "path", "directlighting"):
        ip = ('"integer maxdepth" [%d] ' % maxdepth) + ip
    out.append('SurfaceIntegrator "%s" %s\n' % (integrator, ip))
    if volume_integrator:
        out.append('VolumeIntegrator %s\n' % volume_integrator)
    if count:
        out.append('Accelerator "countaccel" "string inner" ["%s"] %s\n' % (accelerator, accel_params))
    else:
        out.append('Accelerator "%s" %s\n' % (accelerator, accel_params))
    return "".join(out)


def for_product(text: str) -> str:
    """The scene text the product's host library parses: the oracle-side helper plugins of a reference run are unwrapped --
    `Sampler "keyed" "string inner" ["X"] "integer seed" [n] ...` becomes `Sampler "X" "integer seed" [n] ...` (the seed of the
    counter-based RNG is a parameter of the product's own samplers) and `Accelerator "countaccel" "string inner" ["Y"] ...`
    becomes `Accelerator "Y" ...`.  libpbrt_host.so knows neither name."""
    import re
    i = text.find("WorldBegin")                              # the options block only: a world can be a gigabyte of triangles
    head = text if i < 0 else text[:i]
    new = re.sub(r'Sampler\s+"keyed"\s+"string inner"\s*\[\s*"(\w+)"\s*\]', r'Sampler "\1"', head)
    new = re.sub(r'Accelerator\s+"countaccel"\s+"string inner"\s*\[\s*"(\w+)"\s*\]', r'Accelerator "\1"', new)
    if new == head:
        return text
    return new if i < 0 else new + text[i:]


def cornell_scene(soup_tris: int = 0, soup_seed: int = 12345, soup_materials=False,
                  world_kwargs=None, **opts) -> str:
    soup = lcg_soup(soup_tris, soup_seed) if soup_tris else None
    wk = dict(world_kwargs or {})
    return options_block(**opts) + cornell_world(soup=soup, soup_materials=soup_materials, **wk)
