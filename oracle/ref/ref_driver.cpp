// oracle/ref/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Driver that runs the *unmodified* reference renderer (compiled from
// /root/reference by oracle/ref/Makefile into oracle/_ref/) on a pbrt-v1 scene
// file.  The reference's own front end needs flex/bison (core/pbrtlex.l,
// core/pbrtparse.y) and its image writer needs OpenEXR (core/exrio.cpp); neither
// exists here, so this driver
//   * feeds the reference through its public C++ API (core/api.h:29-85) exactly
//     as the grammar actions do (core/pbrtparse.y:294-468), using the repo's own
//     scene parser (pbrt-v1_amd/csrc/host/scene_parser.h) as the tokenizer;
//   * defines the symbols the un-buildable files would have provided:
//     line_num / current_file (pbrtparse.y:31-32), ParseFile (parser.cpp:27),
//     ReadImage / WriteRGBAImage (exrio.cpp:29-96, signatures pbrt.h:234-238).
//     WriteRGBAImage is the float tap: it dumps the pre-quantisation film.
// Nothing from /root/reference is copied; the reference is only #included and
// linked where it lies.
//
// usage: pbrt_ref [--out film.bin] [--quiet] scene.pbrt
// film.bin layout (little endian): char magic[8]="PBRTFILM"; int32 xRes,yRes,
//   totalX,totalY,xOff,yOff; float rgb[3*xRes*yRes]; float alpha[xRes*yRes].
#include "pbrt.h"
#include "api.h"
#include "paramset.h"
#include "color.h"
#include <sys/time.h>
#include <unistd.h>
#include <fcntl.h>

#include "../../pbrt-v1_amd/csrc/host/scene_parser.h"

int line_num = 0;
string current_file;

// ---- shared counters, filled in by the helper plugins (count_accel / keyed_sampler)
extern "C" {
unsigned long long g_ref_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [0]=Intersect calls [1]=IntersectP calls [2]=Intersect hits [3]=IntersectP hits
double g_ref_times[4] = {0, 0, 0, 0};                             // [0]=accel build s  [1]=t(build done)  [2]=t(film written)
double ref_now() { timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + 1e-6 * tv.tv_usec; }
const void *g_keyed_inner_sampler = NULL;                         // set by keyed_sampler.cpp, read by hip_adapter.cpp
unsigned g_keyed_seed = 0;
}

static string g_outPath;

COREDLL bool ParseFile(const char *) { return false; }
COREDLL Spectrum *ReadImage(const string &, int *, int *) { return NULL; }
COREDLL void WriteRGBAImage(const string &name, float *pixels, float *alpha, int XRes, int YRes,
                            int totalXRes, int totalYRes, int xOffset, int yOffset) {
    g_ref_times[2] = ref_now();
    string path = g_outPath.empty() ? name + ".film" : g_outPath;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { Error("ref_driver: cannot write %s", path.c_str()); return; }
    int hdr[6] = {XRes, YRes, totalXRes, totalYRes, xOffset, yOffset};
    fwrite("PBRTFILM", 1, 8, f);
    fwrite(hdr, sizeof(int), 6, f);
    fwrite(pixels, sizeof(float), size_t(3) * XRes * YRes, f);
    fwrite(alpha, sizeof(float), size_t(XRes) * YRes, f);
    fclose(f);
}

namespace {
using pbrthip::ParamList;
using pbrthip::ParamType;

// ParamList -> reference ParamSet, the job of InitParamSet (pbrtparse.y:470-546)
void Fill(ParamSet &ps, const ParamList &pl) {
    for (size_t k = 0; k < pl.size(); ++k) {
        const pbrthip::Param &p = pl[k];
        int n = int(p.nums.size());
        switch (p.type) {
        case ParamType::Float: ps.AddFloat(p.name, p.nums.data(), n); break;
        case ParamType::Int: {
            vector<int> iv(n); for (int i = 0; i < n; ++i) iv[i] = int(p.nums[i]);
            ps.AddInt(p.name, iv.data(), n); break; }
        case ParamType::Bool: {
            int m = int(p.strs.size()); bool *bv = new bool[m ? m : 1];
            for (int i = 0; i < m; ++i) bv[i] = (p.strs[0] == "true");
            ps.AddBool(p.name, bv, m); delete[] bv; break; }
        case ParamType::Point: ps.AddPoint(p.name, (const Point *)p.nums.data(), n / 3); break;
        case ParamType::Vector: ps.AddVector(p.name, (const Vector *)p.nums.data(), n / 3); break;
        case ParamType::Normal: ps.AddNormal(p.name, (const Normal *)p.nums.data(), n / 3); break;
        case ParamType::Color: {
            vector<Spectrum> sv;
            for (int i = 0; i + 2 < n; i += 3) { float c[3] = {p.nums[i], p.nums[i + 1], p.nums[i + 2]}; sv.push_back(Spectrum(c)); }
            ps.AddSpectrum(p.name, sv.data(), int(sv.size())); break; }
        case ParamType::String: ps.AddString(p.name, p.strs.data(), int(p.strs.size())); break;
        case ParamType::Texture: if (p.strs.size() == 1) ps.AddTexture(p.name, p.strs[0]); break;
        }
    }
}

struct RefSink : pbrthip::DirectiveSink {
    void Identity() { pbrtIdentity(); }
    void Translate(float x, float y, float z) { pbrtTranslate(x, y, z); }
    void Rotate(float a, float x, float y, float z) { pbrtRotate(a, x, y, z); }
    void Scale(float x, float y, float z) { pbrtScale(x, y, z); }
    void LookAt(const float v[9]) { pbrtLookAt(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]); }
    void ConcatTransform(const float m[16]) { float t[16]; memcpy(t, m, sizeof t); pbrtConcatTransform(t); }
    void Transform(const float m[16]) { float t[16]; memcpy(t, m, sizeof t); pbrtTransform(t); }
    void CoordinateSystem(const std::string &n) { pbrtCoordinateSystem(n); }
    void CoordSysTransform(const std::string &n) { pbrtCoordSysTransform(n); }
#define FWD(NAME, CALL) void NAME(const std::string &n, const ParamList &p) { ParamSet ps; Fill(ps, p); CALL(n, ps); }
    FWD(PixelFilter, pbrtPixelFilter) FWD(Film, pbrtFilm) FWD(Sampler, pbrtSampler)
    FWD(Accelerator, pbrtAccelerator) FWD(SurfaceIntegrator, pbrtSurfaceIntegrator)
    FWD(VolumeIntegrator, pbrtVolumeIntegrator) FWD(Camera, pbrtCamera) FWD(Material, pbrtMaterial)
    FWD(LightSource, pbrtLightSource) FWD(AreaLightSource, pbrtAreaLightSource) FWD(Shape, pbrtShape)
    FWD(Volume, pbrtVolume)
#undef FWD
    void SearchPath(const std::string &n) { pbrtSearchPath(n); }
    void WorldBegin() { pbrtWorldBegin(); }
    void AttributeBegin() { pbrtAttributeBegin(); }
    void AttributeEnd() { pbrtAttributeEnd(); }
    void TransformBegin() { pbrtTransformBegin(); }
    void TransformEnd() { pbrtTransformEnd(); }
    void Texture(const std::string &name, const std::string &type, const std::string &cls, const ParamList &p) {
        ParamSet ps; Fill(ps, p); pbrtTexture(name, type, cls, ps);
    }
    void ReverseOrientation() { pbrtReverseOrientation(); }
    void ObjectBegin(const std::string &n) { pbrtObjectBegin(n); }
    void ObjectEnd() { pbrtObjectEnd(); }
    void ObjectInstance(const std::string &n) { pbrtObjectInstance(n); }
    void WorldEnd() { pbrtWorldEnd(); }
};
}  // namespace

// ---- the Cornell box of BASELINE.json configs[0] / configs[1] issued through the reference's API (core/api.h:29-85) by hand-written
// calls: no scene text, no tokenizer, no ParamList -- the reference run that does NOT share the product's parser
// (tests/test_parity_chain.py compares it with the run of the equivalent scene file).  Numbers are those of
// pbrt-v1_amd/scenes.py (CORNELL_QUADS, CORNELL_LIGHT); kind: "c1" Whitted 1 spp box filter, "c2" path maxdepth 5 jittered 2x2 mitchell.
static void one_float(ParamSet &ps, const char *n, float v) { ps.AddFloat(n, &v, 1); }
static void one_int(ParamSet &ps, const char *n, int v) { ps.AddInt(n, &v, 1); }
static void one_string(ParamSet &ps, const char *n, const char *v) { string sv(v); ps.AddString(n, &sv, 1); }
static void one_color(ParamSet &ps, const char *n, float r, float g, float b) { float c[3] = {r, g, b}; Spectrum sp(c); ps.AddSpectrum(n, &sp, 1); }
static void quad(const float *v12, const char *mat, float r, float g, float b) {
    pbrtAttributeBegin();
    { ParamSet m; one_color(m, "Kd", r, g, b); pbrtMaterial(mat, m); }
    ParamSet ps; int idx[6] = {0, 1, 2, 0, 2, 3};
    ps.AddInt("indices", idx, 6); ps.AddPoint("P", (const Point *)v12, 4);
    pbrtShape("trianglemesh", ps);
    pbrtAttributeEnd();
}
static bool BuiltinCornell(const string &kind, int res, bool keyed) {
    const bool c2 = kind == "c2";
    pbrtLookAt(278, 273, -800, 278, 273, 0, 0, 1, 0);
    { ParamSet ps; one_float(ps, "fov", 39.3f); pbrtCamera("perspective", ps); }
    { ParamSet ps; one_int(ps, "xresolution", res); one_int(ps, "yresolution", res); one_string(ps, "filename", "out.exr"); pbrtFilm("image", ps); }
    { ParamSet ps; one_int(ps, "xsamples", c2 ? 2 : 1); one_int(ps, "ysamples", c2 ? 2 : 1); bool j = c2; ps.AddBool("jitter", &j, 1);
      if (keyed) { one_string(ps, "inner", "stratified"); one_int(ps, "seed", 0); pbrtSampler("keyed", ps); } else pbrtSampler("stratified", ps); }
    { ParamSet ps; pbrtPixelFilter(c2 ? "mitchell" : "box", ps); }
    { ParamSet ps; one_int(ps, "maxdepth", 5); pbrtSurfaceIntegrator(c2 ? "path" : "whitted", ps); }
    { ParamSet ps; one_string(ps, "inner", "kdtree"); pbrtAccelerator("countaccel", ps); }
    pbrtWorldBegin();
    static const float floor_[12] = {552.8f, 0, 0, 0, 0, 0, 0, 0, 559.2f, 549.6f, 0, 559.2f};
    static const float ceil_[12] = {556, 548.8f, 0, 556, 548.8f, 559.2f, 0, 548.8f, 559.2f, 0, 548.8f, 0};
    static const float back_[12] = {549.6f, 0, 559.2f, 0, 0, 559.2f, 0, 548.8f, 559.2f, 556, 548.8f, 559.2f};
    static const float right_[12] = {0, 0, 559.2f, 0, 0, 0, 0, 548.8f, 0, 0, 548.8f, 559.2f};
    static const float left_[12] = {552.8f, 0, 0, 549.6f, 0, 559.2f, 556, 548.8f, 559.2f, 556, 548.8f, 0};
    static const float light_[12] = {343, 548.7f, 227, 343, 548.7f, 332, 213, 548.7f, 332, 213, 548.7f, 227};
    quad(floor_, "matte", .73f, .73f, .73f); quad(ceil_, "matte", .73f, .73f, .73f); quad(back_, "matte", .73f, .73f, .73f);
    quad(right_, "matte", .12f, .45f, .15f); quad(left_, "matte", .65f, .05f, .05f);
    pbrtAttributeBegin();
    { ParamSet ps; one_color(ps, "L", 17, 12, 4); one_int(ps, "nsamples", 1); pbrtAreaLightSource("area", ps); }
    quad(light_, "matte", 0, 0, 0);
    pbrtAttributeEnd();
    pbrtWorldEnd();
    return true;
}

// ---- grammar scenes: the statements of tests/golden/make_api_fixtures.py's scene texts issued as hand-written pbrt* calls (no scene text, no tokenizer,
// no ParamList in this run).  Each text exercises one corner of the scene language (pbrtlex.l / pbrtparse.y:294-574) that the product's own parser
// must get right; the fixture is the film of THIS run, so the product's tokenizer is checked against something it did not take part in.
static void fv(ParamSet &ps, const char *n, std::initializer_list<float> v) { vector<float> a(v); ps.AddFloat(n, a.data(), int(a.size())); }
static void iv(ParamSet &ps, const char *n, std::initializer_list<int> v) { vector<int> a(v); ps.AddInt(n, a.data(), int(a.size())); }
static void pv(ParamSet &ps, const char *n, std::initializer_list<float> v) { vector<float> a(v); ps.AddPoint(n, (const Point *)a.data(), int(a.size() / 3)); }
static void one_bool(ParamSet &ps, const char *n, bool v) { ps.AddBool(n, &v, 1); }
static void g_sampler(bool keyed, const char *inner, int xs, int ys, bool jitter, int pixelsamples) {
    ParamSet ps;
    if (keyed) { one_string(ps, "inner", inner); one_int(ps, "seed", 0); }
    if (string(inner) == "lowdiscrepancy") one_int(ps, "pixelsamples", pixelsamples);
    else { one_int(ps, "xsamples", xs); one_int(ps, "ysamples", ys); one_bool(ps, "jitter", jitter); }
    pbrtSampler(keyed ? "keyed" : inner, ps);
}
static void g_accel() { ParamSet ps; one_string(ps, "inner", "kdtree"); pbrtAccelerator("countaccel", ps); }
static void g_film(int xr, int yr) { ParamSet ps; one_int(ps, "xresolution", xr); one_int(ps, "yresolution", yr); one_string(ps, "filename", "out.exr"); pbrtFilm("image", ps); }
static void g_point_light(float x, float y, float z, float i) { ParamSet ps; pv(ps, "from", {x, y, z}); one_color(ps, "I", i, i, i); pbrtLightSource("point", ps); }
static void g_quad_shape() { ParamSet ps; iv(ps, "indices", {0, 1, 2, 0, 2, 3}); pv(ps, "P", {-1, -1, 0, 1, -1, 0, 1, 1, 0, -1, 1, 0}); pbrtShape("trianglemesh", ps); }
static void g_matte(float r, float g, float b) { ParamSet ps; one_color(ps, "Kd", r, g, b); pbrtMaterial("matte", ps); }
static bool BuiltinGrammar(const string &kind, bool keyed) {
    if (kind == "g1") {                  // transform stack, named coordinate systems, ReverseOrientation, attribute inheritance
        pbrtLookAt(0, 0, -6, 0, 0, 0, 0, 1, 0);
        { ParamSet ps; one_float(ps, "fov", 40); pbrtCamera("perspective", ps); }
        g_film(32, 32); g_sampler(keyed, "stratified", 1, 1, false, 0);
        { ParamSet ps; pbrtPixelFilter("box", ps); }
        { ParamSet ps; one_int(ps, "maxdepth", 3); pbrtSurfaceIntegrator("whitted", ps); }
        g_accel();
        pbrtWorldBegin();
        g_point_light(0, 4, -4, 60);
        pbrtCoordinateSystem("base");
        pbrtTransformBegin();
        pbrtTranslate(-1, 0.25f, 0); pbrtRotate(30, 0, 1, 0); pbrtScale(0.8f, 1.2f, 1);
        pbrtAttributeBegin(); g_matte(.8f, .2f, .2f); g_quad_shape(); pbrtAttributeEnd();
        pbrtTransformBegin();
        { float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.5f, 1.5f, 0.5f, 1}; pbrtConcatTransform(m); }
        { ParamSet ps; iv(ps, "indices", {0, 1, 2}); pv(ps, "P", {-0.5f, -0.5f, 0, 0.5f, -0.5f, 0, 0, 0.5f, 0}); pbrtShape("trianglemesh", ps); }
        pbrtTransformEnd();
        pbrtTransformEnd();
        { float m[16] = {0.5f, 0, 0, 0, 0, 0.5f, 0, 0, 0, 0, 0.5f, 0, 1.5f, -0.5f, 1, 1}; pbrtTransform(m); }
        pbrtAttributeBegin();
        pbrtReverseOrientation();
        { ParamSet ps; one_color(ps, "Kr", .9f, .9f, .9f); pbrtMaterial("mirror", ps); }
        g_quad_shape();
        pbrtAttributeEnd();
        pbrtIdentity();
        pbrtCoordSysTransform("base");
        pbrtTranslate(0, -1.6f, 0); pbrtRotate(90, 1, 0, 0); pbrtScale(4, 4, 1);
        g_matte(.4f, .4f, .7f);
        g_quad_shape();
        pbrtWorldEnd();
        return true;
    }
    if (kind == "g2") {                  // token rules: comments, number forms, unbracketed single values, line breaks inside parameter lists
        pbrtLookAt(0, 0, -6, 0, 0, 0, 0, 1, 0);
        { ParamSet ps; one_float(ps, "fov", 40); pbrtCamera("perspective", ps); }
        g_film(32, 32); g_sampler(keyed, "stratified", 1, 1, false, 0);
        { ParamSet ps; one_float(ps, "alpha", 1.5f); one_float(ps, "xwidth", 1.5f); one_float(ps, "ywidth", 1.5f); pbrtPixelFilter("gaussian", ps); }
        { ParamSet ps; one_int(ps, "maxdepth", 2); pbrtSurfaceIntegrator("whitted", ps); }
        g_accel();
        pbrtWorldBegin();
        g_point_light(0, 4, -4, 60);
        pbrtAttributeBegin(); g_matte(.5f, .5f, .5f);
        { ParamSet ps; iv(ps, "indices", {0, 1, 2}); pv(ps, "P", {-1.5f, -1, 0, 1.5f, -1, 0, 0, 1.5f, 0}); pbrtShape("trianglemesh", ps); }
        pbrtAttributeEnd();
        pbrtAttributeBegin();
        pbrtTranslate(0, -.5f, -1); pbrtRotate(60, 1, 0, 0);
        { ParamSet ps; one_color(ps, "Kd", .2f, .7f, .3f); one_float(ps, "sigma", 20); pbrtMaterial("matte", ps); }
        { ParamSet ps; iv(ps, "indices", {0, 1, 2, 0, 2, 3}); pv(ps, "P", {-2, -2, 0, 2, -2, 0, 2, 2, 0, -2, 2, 0}); pbrtShape("trianglemesh", ps); }
        pbrtAttributeEnd();
        pbrtWorldEnd();
        return true;
    }
    if (kind == "g3") {                  // Include, constant textures, their attribute scope
        pbrtLookAt(0, 0, -6, 0, 0, 0, 0, 1, 0);
        { ParamSet ps; one_float(ps, "fov", 40); pbrtCamera("perspective", ps); }
        g_film(32, 32); g_sampler(keyed, "stratified", 1, 1, false, 0);
        { ParamSet ps; pbrtPixelFilter("box", ps); }
        { ParamSet ps; one_int(ps, "maxdepth", 2); pbrtSurfaceIntegrator("whitted", ps); }
        g_accel();
        pbrtWorldBegin();
        g_point_light(0, 4, -4, 60);
        { ParamSet ps; one_color(ps, "value", .7f, .3f, .1f); pbrtTexture("rust", "color", "constant", ps); }
        { ParamSet ps; one_float(ps, "value", 30); pbrtTexture("rough", "float", "constant", ps); }
        pbrtAttributeBegin();
        { ParamSet ps; ps.AddTexture("Kd", "rust"); ps.AddTexture("sigma", "rough"); pbrtMaterial("matte", ps); }
        pbrtTranslate(-1.2f, 0, 0);
        g_quad_shape();                                    // the included file's statement
        pbrtAttributeEnd();
        pbrtAttributeBegin();
        { ParamSet ps; one_color(ps, "value", .1f, .3f, .8f); pbrtTexture("rust", "color", "constant", ps); }      // redefined inside the attribute block
        { ParamSet ps; ps.AddTexture("Kd", "rust"); pbrtMaterial("matte", ps); }
        pbrtTranslate(1.2f, 0, 0);
        g_quad_shape();                                    // the same file included again
        pbrtAttributeEnd();
        { ParamSet ps; ps.AddTexture("Kd", "rust"); pbrtMaterial("matte", ps); }                                    // the outer definition again
        pbrtTranslate(0, -1.6f, 0); pbrtRotate(90, 1, 0, 0); pbrtScale(4, 4, 1);
        g_quad_shape();
        pbrtWorldEnd();
        return true;
    }
    if (kind == "g4") {                  // parameter typing: integers written as floats are truncated, bools are strings, area light under a transform
        pbrtLookAt(0, 0, -6, 0, 0, 0, 0, 1, 0);
        { ParamSet ps; one_float(ps, "fov", 40); pbrtCamera("perspective", ps); }
        g_film(32, 32); g_sampler(keyed, "stratified", 2, 2, true, 0);
        { ParamSet ps; one_float(ps, "B", .3f); one_float(ps, "C", .35f); pbrtPixelFilter("mitchell", ps); }
        { ParamSet ps; one_string(ps, "strategy", "all"); one_int(ps, "maxdepth", 2); pbrtSurfaceIntegrator("directlighting", ps); }
        g_accel();
        pbrtWorldBegin();
        pbrtAttributeBegin();
        { ParamSet ps; one_color(ps, "L", 20, 18, 15); one_int(ps, "nsamples", 2); pbrtAreaLightSource("area", ps); }
        pbrtTranslate(0, 2.5f, 0); pbrtRotate(90, 1, 0, 0);
        g_quad_shape();
        pbrtAttributeEnd();
        pbrtAttributeBegin(); g_matte(.7f, .7f, .7f); pbrtTranslate(0, -.3f, .5f); pbrtRotate(-20, 0, 1, 0); g_quad_shape(); pbrtAttributeEnd();
        g_matte(.3f, .6f, .3f);
        pbrtTranslate(0, -1.6f, 0); pbrtRotate(90, 1, 0, 0); pbrtScale(4, 4, 1);
        g_quad_shape();
        pbrtWorldEnd();
        return true;
    }
    if (kind == "g5") {                  // factory defaults, an unused parameter, camera / film options, a light placed by the CTM only
        pbrtLookAt(0, 0, -6, 0, 0, 0, 0, 1, 0);
        { ParamSet ps; one_float(ps, "fov", 35); one_float(ps, "lensradius", .05f); one_float(ps, "focaldistance", 6); one_float(ps, "frameaspectratio", 1);
          fv(ps, "screenwindow", {-1, 1, -1, 1}); one_float(ps, "hither", .01f); one_float(ps, "yon", 100); pbrtCamera("perspective", ps); }
        { ParamSet ps; one_int(ps, "xresolution", 40); one_int(ps, "yresolution", 30); one_string(ps, "filename", "out.exr"); fv(ps, "cropwindow", {.1f, .9f, .2f, 1}); pbrtFilm("image", ps); }
        g_sampler(keyed, "lowdiscrepancy", 0, 0, false, 4);
        { ParamSet ps; pbrtPixelFilter("triangle", ps); }
        { ParamSet ps; one_string(ps, "strategy", "one"); one_float(ps, "bogus", 1); pbrtSurfaceIntegrator("directlighting", ps); }
        g_accel();
        pbrtWorldBegin();
        pbrtTransformBegin(); pbrtTranslate(0, 3, -3);
        { ParamSet ps; one_color(ps, "I", 50, 50, 50); pbrtLightSource("point", ps); }       // "from" defaults to the origin: the CTM places the light
        pbrtTransformEnd();
        pbrtAttributeBegin(); { ParamSet ps; pbrtMaterial("mirror", ps); } pbrtTranslate(-1.1f, 0, .5f); pbrtRotate(25, 0, 1, 0); g_quad_shape(); pbrtAttributeEnd();
        pbrtAttributeBegin(); { ParamSet ps; one_color(ps, "Kd", .3f, .3f, .6f); pbrtMaterial("plastic", ps); } pbrtTranslate(1.1f, 0, 0); g_quad_shape(); pbrtAttributeEnd();
        pbrtTranslate(0, -1.6f, 0); pbrtRotate(90, 1, 0, 0); pbrtScale(4, 4, 1);
        g_quad_shape();                                                                           // no Material statement at all: the default matte
        pbrtWorldEnd();
        return true;
    }
    fprintf(stderr, "unknown builtin scene %s\n", kind.c_str());
    return false;
}

int main(int argc, char **argv) {
    bool quiet = false; string scene, builtin; int builtin_res = 64; bool builtin_keyed = false;
    for (int i = 1; i < argc; ++i) {
        string a = argv[i];
        if (a == "--out" && i + 1 < argc) g_outPath = argv[++i];
        else if (a == "--quiet") quiet = true;
        else if (a == "--builtin" && i + 2 < argc) { builtin = argv[++i]; builtin_res = atoi(argv[++i]); scene = "<builtin>"; }
        else if (a == "--keyed-sampler") builtin_keyed = true;
        else scene = a;
    }
    if (scene.empty()) { fprintf(stderr, "usage: %s [--out film.bin] [--quiet] scene.pbrt\n", argv[0]); return 2; }
    int savedOut = -1;
    if (quiet) {  // progress bar + StatsPrint go to stdout (util.cpp:396-448, api.cpp:479)
        fflush(stdout); savedOut = dup(1);
        int nul = open("/dev/null", O_WRONLY); dup2(nul, 1); close(nul);
    }
    double t0 = ref_now();
    pbrtInit();
    RefSink sink;
    pbrthip::SceneParser parser(sink);
    current_file = scene;
    bool ok = builtin.empty() ? parser.ParseFile(scene) : (builtin[0] == 'g' ? BuiltinGrammar(builtin, builtin_keyed) : BuiltinCornell(builtin, builtin_res, builtin_keyed));
    pbrtCleanup();
    double t1 = ref_now();
    if (quiet) { fflush(stdout); dup2(savedOut, 1); close(savedOut); }
    double render_s = (g_ref_times[1] > 0 && g_ref_times[2] > 0) ? g_ref_times[2] - g_ref_times[1] : -1.;
    printf("{\"ok\": %s, \"total_s\": %.6f, \"accel_build_s\": %.6f, \"render_s\": %.6f, "
           "\"closest_rays\": %llu, \"any_rays\": %llu, \"closest_hits\": %llu, \"any_hits\": %llu}\n",
           ok ? "true" : "false", t1 - t0, g_ref_times[0], render_s,
           g_ref_counters[0], g_ref_counters[1], g_ref_counters[2], g_ref_counters[3]);
    return ok ? 0 : 1;
}
