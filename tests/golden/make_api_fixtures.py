"""api_g*.npz: the scene LANGUAGE pinned without the shared tokenizer.  Authoring container only:

    python tests/golden/make_api_fixtures.py

Every reference run that produces a film fixture tokenizes its scene file with the product's own parser (oracle/ref/ref_driver.cpp: flex / bison
are absent from the image), so a tokenizer bug would be common mode.  Here each scene text below is ALSO written out by hand as pbrt* API calls
(ref_driver.cpp BuiltinGrammar "g1" .. "g5": no text, no tokenizer, no ParamList in that run); the fixture holds the film of the API-driven run,
and -- asserted here -- the run of the text gives the same film bit for bit.  tests/test_parity_chain.py then has the oracle and the device parse
the TEXT and reproduce the API run's film.

  g1  transform stack: Translate / Rotate / Scale / ConcatTransform / Transform / Identity, TransformBegin / End nesting, CoordinateSystem +
      CoordSysTransform, ReverseOrientation, material inheritance across AttributeBegin / End            (pbrtparse.y:294-420, api.cpp:109-257)
  g2  token rules: comments, number forms (4., .15e1, +1.5, 6e1), unbracketed single values, line breaks inside parameter lists, several
      statements on one line                                                                          (pbrtlex.l:95-178)
  g3  Include (twice, relative to the including file), constant float / color textures and their attribute scope   (pbrtlex.l:120-150, api.cpp:330-390)
  g4  parameter typing: integers written as floats are truncated, bools are the strings "true" / "false", an area light under a transform
                                                                                                      (pbrtparse.y:470-574)
  g5  factory defaults (materials without parameters, no Material statement, a point light without "from"), an unused parameter (Warning),
      camera / film options (lens, screenwindow, cropwindow), the LD sampler                          (dynload.cpp:112-260)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
REF = g.load_ref_runner()
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chain")

QUAD = 'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -1 0 1 -1 0 1 1 0 -1 1 0]\n'
KEYED = 'Sampler "keyed" "string inner" ["%s"] "integer seed" [0] %s\n'
ACCEL = 'Accelerator "countaccel" "string inner" ["kdtree"]\n'
HEAD = 'LookAt 0 0 -6  0 0 0  0 1 0\nCamera "perspective" "float fov" [40]\nFilm "image" "integer xresolution" [32] "integer yresolution" [32] "string filename" ["out.exr"]\n'
FLOOR = 'Translate 0 -1.6 0\nRotate 90 1 0 0\nScale 4 4 1\n'

SCENES = {}
SCENES["g1"] = (HEAD + KEYED % ("stratified", '"integer xsamples" [1] "integer ysamples" [1] "bool jitter" ["false"]') + 'PixelFilter "box"\n'
    'SurfaceIntegrator "whitted" "integer maxdepth" [3]\n' + ACCEL + 'WorldBegin\n'
    'LightSource "point" "point from" [0 4 -4] "color I" [60 60 60]\n'
    'CoordinateSystem "base"\n'
    'TransformBegin\n  Translate -1 0.25 0\n  Rotate 30 0 1 0\n  Scale 0.8 1.2 1\n'
    '  AttributeBegin\n    Material "matte" "color Kd" [.8 .2 .2]\n    ' + QUAD + '  AttributeEnd\n'
    '  TransformBegin\n    ConcatTransform [1 0 0 0  0 1 0 0  0 0 1 0  0.5 1.5 0.5 1]\n'
    '    Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-0.5 -0.5 0 0.5 -0.5 0 0 0.5 0]\n  TransformEnd\nTransformEnd\n'
    'Transform [0.5 0 0 0  0 0.5 0 0  0 0 0.5 0  1.5 -0.5 1 1]\n'
    'AttributeBegin\n  ReverseOrientation\n  Material "mirror" "color Kr" [.9 .9 .9]\n  ' + QUAD + 'AttributeEnd\n'
    'Identity\nCoordSysTransform "base"\n' + FLOOR + 'Material "matte" "color Kd" [.4 .4 .7]\n' + QUAD + 'WorldEnd\n', {})
SCENES["g2"] = ('# api_g2: the token rules of pbrtlex.l\nLookAt 0 0 -6 0 0 0 0 1 0 # a trailing comment\n'
    'Camera "perspective" "float fov" 40\n'
    'Film "image" "integer xresolution" 32 "integer yresolution" [ 32 ]\n     "string filename" "out.exr"\n'
    + KEYED % ("stratified", '"integer xsamples" [1] "integer ysamples" [1] "bool jitter" ["false"]') +
    'PixelFilter "gaussian" "float alpha" [1.5e0] "float xwidth" [+1.5] "float ywidth" [ .15e1 ]\n'
    'SurfaceIntegrator "whitted" "integer maxdepth" [2]\n' + ACCEL + 'WorldBegin\n'
    'LightSource "point" "point from" [ 0 4. -4 ]\n   "color I" [6e1 60 60.0]\n'
    'AttributeBegin Material "matte" "color Kd" [ .5 .5\n  .5 ] Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1.5 -1 0 1.5 -1 0 0 1.5 0] AttributeEnd # three statements, one line\n'
    'AttributeBegin\n  Translate 0 -.5 -1 Rotate 60 1 0 0     # two on a line\n'
    '  Material "matte" "color Kd" [.2 .7 .3] "float sigma" 20\n'
    '  Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ -2 -2 0   2 -2 0\n      2 2 0   -2 2 0 ]\nAttributeEnd\nWorldEnd\n', {})
SCENES["g3"] = (HEAD + KEYED % ("stratified", '"integer xsamples" [1] "integer ysamples" [1] "bool jitter" ["false"]') + 'PixelFilter "box"\n'
    'SurfaceIntegrator "whitted" "integer maxdepth" [2]\n' + ACCEL + 'WorldBegin\n'
    'LightSource "point" "point from" [0 4 -4] "color I" [60 60 60]\n'
    'Texture "rust" "color" "constant" "color value" [.7 .3 .1]\nTexture "rough" "float" "constant" "float value" [30]\n'
    'AttributeBegin\n  Material "matte" "texture Kd" "rust" "texture sigma" "rough"\n  Translate -1.2 0 0\n  Include "parts/quad.pbrt"\nAttributeEnd\n'
    'AttributeBegin\n  Texture "rust" "color" "constant" "color value" [.1 .3 .8]   # scoped: the outer "rust" is back after AttributeEnd\n'
    '  Material "matte" "texture Kd" "rust"\n  Translate 1.2 0 0\n  Include "parts/quad.pbrt"\nAttributeEnd\n'
    'Material "matte" "texture Kd" "rust"\n' + FLOOR + QUAD + 'WorldEnd\n', {"parts/quad.pbrt": "# included twice\n" + QUAD})
SCENES["g4"] = ('LookAt 0 0 -6  0 0 0  0 1 0\nCamera "perspective" "float fov" [40]\n'
    'Film "image" "integer xresolution" [32.9] "integer yresolution" [32.2] "string filename" ["out.exr"]\n'
    + KEYED % ("stratified", '"integer xsamples" [2.7] "integer ysamples" [2] "bool jitter" ["true"]') +
    'PixelFilter "mitchell" "float B" [.3] "float C" [.35]\n'
    'SurfaceIntegrator "directlighting" "string strategy" ["all"] "integer maxdepth" [2]\n' + ACCEL + 'WorldBegin\n'
    'AttributeBegin\n  AreaLightSource "area" "color L" [20 18 15] "integer nsamples" [2.5]\n  Translate 0 2.5 0\n  Rotate 90 1 0 0\n'
    '  Shape "trianglemesh" "integer indices" [0.0 1.0 2.0 0 2 3.0] "point P" [-1 -1 0 1 -1 0 1 1 0 -1 1 0]\nAttributeEnd\n'
    'AttributeBegin\n  Material "matte" "color Kd" [.7 .7 .7]\n  Translate 0 -.3 .5\n  Rotate -20 0 1 0\n  ' + QUAD + 'AttributeEnd\n'
    'Material "matte" "color Kd" [.3 .6 .3]\n' + FLOOR + QUAD + 'WorldEnd\n', {})
SCENES["g5"] = ('LookAt 0 0 -6  0 0 0  0 1 0\n'
    'Camera "perspective" "float fov" [35] "float lensradius" [.05] "float focaldistance" [6] "float frameaspectratio" [1] "float screenwindow" [-1 1 -1 1] "float hither" [.01] "float yon" [100]\n'
    'Film "image" "integer xresolution" [40] "integer yresolution" [30] "string filename" ["out.exr"] "float cropwindow" [.1 .9 .2 1]\n'
    + KEYED % ("lowdiscrepancy", '"integer pixelsamples" [4]') + 'PixelFilter "triangle"\n'
    'SurfaceIntegrator "directlighting" "string strategy" ["one"] "float bogus" [1]\n' + ACCEL + 'WorldBegin\n'
    'TransformBegin\n  Translate 0 3 -3\n  LightSource "point" "color I" [50 50 50]\nTransformEnd\n'
    'AttributeBegin\n  Material "mirror"\n  Translate -1.1 0 .5\n  Rotate 25 0 1 0\n  ' + QUAD + 'AttributeEnd\n'
    'AttributeBegin\n  Material "plastic" "color Kd" [.3 .3 .6]\n  Translate 1.1 0 0\n  ' + QUAD + 'AttributeEnd\n'
    + FLOOR + QUAD + 'WorldEnd\n', {})


def main():
    import tempfile
    for kind, (text, files) in SCENES.items():
        d = tempfile.mkdtemp(prefix="apifix_")
        for rel, body in files.items():
            os.makedirs(os.path.dirname(os.path.join(d, rel)), exist_ok=True)
            open(os.path.join(d, rel), "w").write(body)
        rgb_file, alpha_file, st = REF.run_reference(text, keyed=True, workdir=d)                 # the text, through the (shared) tokenizer
        rgb_api, alpha_api, st_api = REF.run_reference(("builtin", kind, 0), keyed=True)          # the hand-written API calls
        assert rgb_api.shape == rgb_file.shape, (kind, rgb_api.shape, rgb_file.shape)
        assert np.array_equal(rgb_file, rgb_api) and np.array_equal(alpha_file, alpha_api), (kind, float(np.abs(rgb_file - rgb_api).max()))
        assert st["closest_rays"] == st_api["closest_rays"] and st["any_rays"] == st_api["any_rays"], kind
        assert float(rgb_api.mean()) > 0.01 and float((rgb_api.sum(-1) > 0).mean()) > 0.3, (kind, "the picture is (nearly) empty")
        np.savez_compressed(os.path.join(OUT, "api_%s.npz" % kind), scene=np.array(text), rgb=rgb_api, alpha=alpha_api,
                            stats=np.array(json.dumps({k: v for k, v in st_api.items() if k != "stats"})),
                            files=np.array(json.dumps(files)))
        print("api_" + kind, rgb_api.shape, "mean %.5f" % float(rgb_api.mean()), "lit %.2f" % float((rgb_api.sum(-1) > 0).mean()), st_api["closest_rays"], st_api["any_rays"])


if __name__ == "__main__":
    main()
