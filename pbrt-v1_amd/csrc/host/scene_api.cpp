// scene_api.cpp -- see scene_api.h.  Citations are to the reference tree (/root/reference).
#include "scene_api.h"
#include "../../../include/pbrt_hip_desc.h"
#include "exr_io.h"
#include "../../../include/pbrt_hip_plugin.h"
#include <dlfcn.h>
#include <unistd.h>
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>

namespace pbrthip {

// ------------------------------------------------------------------ diagnostics (core/util.cpp:36-97)
static int g_warnings = 0, g_errors = 0; static bool g_quiet = false;
int WarningCount() { return g_warnings; }
int ErrorCount() { return g_errors; }
void ResetDiagnostics() { g_warnings = g_errors = 0; }
void SetQuiet(bool q) { g_quiet = q; }
void Warning(const char *fmt, ...) {
    ++g_warnings; if (g_quiet) return;
    va_list a; va_start(a, fmt); std::fprintf(stderr, "Warning: "); std::vfprintf(stderr, fmt, a); std::fprintf(stderr, "\n"); va_end(a);
}
void Error(const char *fmt, ...) {
    ++g_errors; if (g_quiet) return;
    va_list a; va_start(a, fmt); std::fprintf(stderr, "Error: "); std::vfprintf(stderr, fmt, a); std::fprintf(stderr, "\n"); va_end(a);
}

// ------------------------------------------------------------------ filters (filters/*.cpp)
float Filter::Evaluate(float x, float y) const {
    if (name == "box") return 1.f;                                                  // box.cpp:36-38
    if (name == "triangle")                                                         // triangle.cpp:36-39
        return std::fmax(0.f, xWidth - std::fabs(x)) * std::fmax(0.f, yWidth - std::fabs(y));
    if (name == "gaussian") {                                                       // gaussian.cpp:29-50
        const float alpha = p0, expX = p1, expY = p2;
        float gx = std::fmax(0.f, float(expf(-alpha * x * x) - expX));
        float gy = std::fmax(0.f, float(expf(-alpha * y * y) - expY));
        return gx * gy;
    }
    if (name == "mitchell") {                                                       // mitchell.cpp:34-51
        const float B = p0, C = p1;
        auto m1 = [B, C](float v) {
            v = std::fabs(2.f * v);
            if (v > 1.f) return ((-B - 6 * C) * v * v * v + (6 * B + 30 * C) * v * v + (-12 * B - 48 * C) * v + (8 * B + 24 * C)) * (1.f / 6.f);
            return ((12 - 9 * B - 6 * C) * v * v * v + (-18 + 12 * B + 6 * C) * v * v + (6 - 2 * B)) * (1.f / 6.f);
        };
        return m1(x * invXWidth) * m1(y * invYWidth);
    }
    if (name == "sinc") {                                                           // sinc.cpp:41-53
        const float tau = p0;
        auto s1 = [tau](float v) {
            v = std::fabs(v);
            if (v < 1e-5) return 1.f;
            if (v > 1.) return 0.f;
            v *= 3.14159265358979323846f;
            float sinc = sinf(v * tau) / (v * tau);
            float lanczos = sinf(v) / v;
            return sinc * lanczos;
        };
        return s1(x * invXWidth) * s1(y * invYWidth);
    }
    return 1.f;
}

Filter MakeFilter(const std::string &name, const ParamSet &ps, bool *ok) {
    Filter f; f.name = name; *ok = true;
    float xw, yw;
    if (name == "box") { xw = ps.FindOneFloat("xwidth", .5f); yw = ps.FindOneFloat("ywidth", .5f); }
    else if (name == "triangle") { xw = ps.FindOneFloat("xwidth", 2.); yw = ps.FindOneFloat("ywidth", 2.); }
    else if (name == "gaussian") {
        xw = ps.FindOneFloat("xwidth", 2.); yw = ps.FindOneFloat("ywidth", 2.);
        f.p0 = ps.FindOneFloat("alpha", 2.f);
        f.p1 = expf(-f.p0 * xw * xw); f.p2 = expf(-f.p0 * yw * yw);
    } else if (name == "mitchell") {
        xw = ps.FindOneFloat("xwidth", 2.); yw = ps.FindOneFloat("ywidth", 2.);
        f.p0 = ps.FindOneFloat("B", 1.f / 3.f); f.p1 = ps.FindOneFloat("C", 1.f / 3.f);
    } else if (name == "sinc") {
        xw = ps.FindOneFloat("xwidth", 4.); yw = ps.FindOneFloat("ywidth", 4.);
        f.p0 = ps.FindOneFloat("tau", 3.f);
    } else { Error("Unable to load plugin \"%s\" (pixel filter)", name.c_str()); *ok = false; xw = yw = .5f; f.name = "box"; }
    f.xWidth = xw; f.yWidth = yw; f.invXWidth = 1.f / xw; f.invYWidth = 1.f / yw;
    ps.ReportUnused();
    return f;
}

// ------------------------------------------------------------------ film (film/image.cpp:69-101,148-156,213-233)
static int ceil2int(double v) { return int(std::ceil(v)); }
static int floor2int(double v) { return int(std::floor(v)); }

Film MakeFilm(const std::string &name, const ParamSet &ps, const Filter &filt, bool *ok) {
    Film f; *ok = true;
    if (name != "image") { Error("Unable to load plugin \"%s\" (film)", name.c_str()); *ok = false; }
    f.filename = ps.FindOneString("filename", "pbrt.exr");
    f.premultiplyAlpha = ps.FindOneBool("premultiplyalpha", true);
    f.xResolution = ps.FindOneInt("xresolution", 640);
    f.yResolution = ps.FindOneInt("yresolution", 480);
    float crop[4] = {0, 1, 0, 1};
    int cwi; const float *cr = ps.FindFloat("cropwindow", &cwi);
    auto clamp01 = [](float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); };
    if (cr && cwi == 4) {
        crop[0] = clamp01(std::fmin(cr[0], cr[1])); crop[1] = clamp01(std::fmax(cr[0], cr[1]));
        crop[2] = clamp01(std::fmin(cr[2], cr[3])); crop[3] = clamp01(std::fmax(cr[2], cr[3]));
    }
    f.writeFrequency = ps.FindOneInt("writefrequency", -1);
    std::memcpy(f.cropWindow, crop, sizeof crop);
    f.xPixelStart = ceil2int(f.xResolution * crop[0]);
    f.xPixelCount = std::max(1, ceil2int(f.xResolution * crop[1]) - f.xPixelStart);
    f.yPixelStart = ceil2int(f.yResolution * crop[2]);
    f.yPixelCount = std::max(1, ceil2int(f.yResolution * crop[3]) - f.yPixelStart);
    f.filter = filt;
    float *ftp = f.filterTable;
    for (int y = 0; y < 16; ++y) {
        float fy = ((float)y + .5f) * filt.yWidth / 16;
        for (int x = 0; x < 16; ++x) {
            float fx = ((float)x + .5f) * filt.xWidth / 16;
            *ftp++ = filt.Evaluate(fx, fy);
        }
    }
    ps.ReportUnused();
    return f;
}
void Film::GetSampleExtent(int *xs, int *xe, int *ys, int *ye) const {
    *xs = floor2int(xPixelStart + .5f - filter.xWidth);
    *xe = floor2int(xPixelStart + .5f + xPixelCount + filter.xWidth);
    *ys = floor2int(yPixelStart + .5f - filter.yWidth);
    *ye = floor2int(yPixelStart + .5f + yPixelCount + filter.yWidth);
}

// ------------------------------------------------------------------ plugin loader (core/dynload.cpp:41-61, 462-514)
// `Kind "name"` first looks for name.so / libname.so along the search path and calls its extern "C" Create<Kind> factory
// (include/pbrt_hip_plugin.h); the plugins of the hot path are compiled in and serve as the fall-back.
static std::vector<std::string> &pluginDirs() { static std::vector<std::string> d; return d; }
static void addPluginPath(const std::string &path) {         // colon-separated, like PBRT_SEARCHPATH (dynload.cpp:86-111 UpdatePluginPath)
    size_t a = 0;
    while (a <= path.size()) {
        size_t b = path.find(':', a); if (b == std::string::npos) b = path.size();
        if (b > a) pluginDirs().push_back(path.substr(a, b - a));
        a = b + 1;
    }
}
// The factories carry product-specific names (PbrtHipCreate*, not the reference's Create*): the descriptors they fill are another ABI than the
// reference's `Create*(const ParamSet &, ...)` returning C++ objects, and a SearchPath that points at a pbrt-v1 install holds stratified.so,
// kdtree.so, ... of exactly the names a scene asks for.  A shared object that does not load here (a reference plugin: its core symbols
// are unresolved) or lacks the product's factory is not an error: a warning, then the compiled-in plugin answers.  PBRT_SEARCHPATH is the
// reference's install and is not searched; PBRT_HIP_PLUGIN_PATH and the scene's SearchPath directives are.
static void *findPluginSymbol(const std::string &name, const char *symbol) {
    static bool env_done = false;
    if (!env_done) {
        env_done = true;
        if (const char *e = std::getenv("PBRT_HIP_PLUGIN_PATH")) addPluginPath(e);
    }
    if (name.empty() || name.find('/') != std::string::npos) return nullptr;
    static std::map<std::string, void *> handles;              // a shared object is opened once per process, as the reference's Plugin cache does (nullptr: did not load)
    for (const std::string &dir : pluginDirs())
        for (const char *prefix : {"", "lib"}) {
            const std::string file = dir + "/" + prefix + name + ".so";
            void *h = nullptr;
            auto it = handles.find(file);
            if (it != handles.end()) h = it->second;
            else {
                if (access(file.c_str(), R_OK) != 0) continue;
                h = dlopen(file.c_str(), RTLD_NOW | RTLD_LOCAL);
                if (!h) Warning("Plugin \"%s\" does not load here (%s); using the built-in \"%s\" if there is one", file.c_str(), dlerror(), name.c_str());
                handles[file] = h;
            }
            if (!h) continue;
            // the ABI the object was built against (include/pbrt_hip_plugin.h PBRT_HIP_PLUGIN_ABI); an object without the symbol predates the check
            if (void *abi = dlsym(h, "PbrtHipPluginAbi")) {
                const int v = reinterpret_cast<int (*)()>(abi)();
                if (v != PBRT_HIP_PLUGIN_ABI) { Warning("Plugin \"%s\" was built for plugin ABI %d, this library speaks %d; ignored", file.c_str(), v, PBRT_HIP_PLUGIN_ABI); continue; }
            }
            if (void *sym = dlsym(h, symbol)) return sym;
        }
    return nullptr;
}
namespace {
struct CParams { const ParamSet *ps; mutable std::deque<std::string> strings; };      // deque: a later find_string must not move the strings handed out before
const CParams *cp(const PbrtHipParams *p) { return reinterpret_cast<const CParams *>(p); }
const PbrtHipParamsApi kParamsApi = {
    [](const PbrtHipParams *p, const char *n, int d) { return cp(p)->ps->FindOneInt(n, d); },
    [](const PbrtHipParams *p, const char *n, float d) { return cp(p)->ps->FindOneFloat(n, d); },
    [](const PbrtHipParams *p, const char *n, int d) { return int(cp(p)->ps->FindOneBool(n, d != 0)); },
    [](const PbrtHipParams *p, const char *n, const char *d) -> const char * {
        cp(p)->strings.push_back(cp(p)->ps->FindOneString(n, d ? d : "")); return cp(p)->strings.back().c_str(); }};
}  // namespace

// ------------------------------------------------------------------ samplers
Sampler MakeSampler(const std::string &nameIn, const ParamSet &ps, const Film &, bool *ok) {
    Sampler s; *ok = true; s.seed = 0; s.pixelsamples = 4; s.xsamples = s.ysamples = 2; s.jitter = true;
    const std::string &name = nameIn;
    if (void *sym = findPluginSymbol(name, "PbrtHipCreateSampler")) {                      // dynload.cpp:230-245 MakeSampler
        CParams c{&ps}; PbrtHipSampler o{};
        if (reinterpret_cast<PbrtHipCreateSamplerFn>(sym)(reinterpret_cast<const PbrtHipParams *>(&c), &kParamsApi, &o) != 0 ||
            o.kind < RT_SAMPLER_STRATIFIED || o.kind > RT_SAMPLER_RANDOM) { Error("Unable to load plugin \"%s\" (sampler)", name.c_str()); *ok = false; s.kind = RT_SAMPLER_STRATIFIED; }
        else { s.kind = o.kind; s.xsamples = o.xsamples; s.ysamples = o.ysamples; s.jitter = o.jitter != 0; s.pixelsamples = o.pixelsamples; s.seed = o.seed; }
        ps.ReportUnused();
        return s;
    }
    // extension: "integer seed" selects the stream of this renderer's counter-based RNG; the reference's samplers draw
    // from one global MT19937 and have no such parameter
    s.seed = unsigned(ps.FindOneInt("seed", 0));
    if (name == "stratified") {                                                     // stratified.cpp:132-141
        s.kind = RT_SAMPLER_STRATIFIED;
        s.jitter = ps.FindOneBool("jitter", true);
        s.xsamples = ps.FindOneInt("xsamples", 2); s.ysamples = ps.FindOneInt("ysamples", 2);
    } else if (name == "lowdiscrepancy") {                                          // lowdiscrepancy.cpp:129-136
        s.kind = RT_SAMPLER_LOWDISCREPANCY; s.pixelsamples = ps.FindOneInt("pixelsamples", 4);
        if (s.pixelsamples & (s.pixelsamples - 1)) {                                // LDSampler ctor lowdiscrepancy.cpp:62-66
            Warning("Pixel samples being rounded up to power of 2");
            unsigned v = unsigned(s.pixelsamples); v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; s.pixelsamples = int(v + 1);
        }
    } else if (name == "random") {                                                  // random.cpp:118-126
        s.kind = RT_SAMPLER_RANDOM; s.xsamples = ps.FindOneInt("xsamples", 2); s.ysamples = ps.FindOneInt("ysamples", 2);
    } else { Error("Unable to load plugin \"%s\" (sampler)", name.c_str()); *ok = false; s.kind = RT_SAMPLER_STRATIFIED; }
    ps.ReportUnused();
    return s;
}

// ------------------------------------------------------------------ integrators
SurfaceIntegrator MakeSurfaceIntegrator(const std::string &name, const ParamSet &ps, bool *ok) {
    SurfaceIntegrator si; *ok = true; si.strategy = RT_STRATEGY_ALL;
    if (void *sym = findPluginSymbol(name, "PbrtHipCreateSurfaceIntegrator")) {            // dynload.cpp:185-199 MakeSurfaceIntegrator
        CParams c{&ps}; PbrtHipSurfaceIntegrator o{};
        if (reinterpret_cast<PbrtHipCreateSurfaceIntegratorFn>(sym)(reinterpret_cast<const PbrtHipParams *>(&c), &kParamsApi, &o) != 0 ||
            o.kind < RT_INTEGRATOR_WHITTED || o.kind > RT_INTEGRATOR_PATH || (o.strategy < RT_STRATEGY_ALL || o.strategy > RT_STRATEGY_WEIGHTED)) {
            Error("Unable to load plugin \"%s\" (surface integrator)", name.c_str()); *ok = false; si.kind = RT_INTEGRATOR_WHITTED; si.maxDepth = 5;
        } else { si.kind = o.kind; si.maxDepth = o.max_depth; si.strategy = o.strategy; }
        ps.ReportUnused();
        return si;
    }
    si.maxDepth = ps.FindOneInt("maxdepth", 5);              // whitted.cpp:143, directlighting.cpp:196, path.cpp:147
    if (name == "whitted") si.kind = RT_INTEGRATOR_WHITTED;
    else if (name == "path") si.kind = RT_INTEGRATOR_PATH;
    else if (name == "directlighting") {                     // directlighting.cpp:195-208
        si.kind = RT_INTEGRATOR_DIRECT;
        std::string st = ps.FindOneString("strategy", "all");
        if (st == "one") si.strategy = RT_STRATEGY_ONE;
        else if (st == "all") si.strategy = RT_STRATEGY_ALL;
        else if (st == "weighted") si.strategy = RT_STRATEGY_WEIGHTED;   // WeightedSampleOneLight transport.cpp:71-122: rt_render runs it on one shard (include/pbrt_hip.h)
        else { Warning("Strategy \"%s\" for direct lighting unknown. Using \"all\".", st.c_str()); si.strategy = RT_STRATEGY_ALL; }
    } else { Error("Unable to load plugin \"%s\" (surface integrator)", name.c_str()); *ok = false; si.kind = RT_INTEGRATOR_WHITTED; }
    ps.ReportUnused();
    return si;
}
VolumeIntegrator MakeVolumeIntegrator(const std::string &name, const ParamSet &ps, bool *ok) {
    VolumeIntegrator vi; *ok = true;
    if (void *sym = findPluginSymbol(name, "PbrtHipCreateVolumeIntegrator")) {             // dynload.cpp:200-214 MakeVolumeIntegrator
        CParams c{&ps}; PbrtHipVolumeIntegrator o{};
        if (reinterpret_cast<PbrtHipCreateVolumeIntegratorFn>(sym)(reinterpret_cast<const PbrtHipParams *>(&c), &kParamsApi, &o) != 0 ||
            (o.kind != RT_VOLUME_EMISSION && o.kind != RT_VOLUME_SINGLE)) { Error("Unable to load plugin \"%s\" (volume integrator)", name.c_str()); *ok = false; vi.kind = RT_VOLUME_EMISSION; vi.stepSize = 1.f; }
        else { vi.kind = o.kind; vi.stepSize = o.step_size; }
        ps.ReportUnused();
        return vi;
    }
    vi.stepSize = ps.FindOneFloat("stepsize", 1.f);          // emission.cpp:97, single.cpp:118
    if (name == "emission") vi.kind = RT_VOLUME_EMISSION;
    else if (name == "single") vi.kind = RT_VOLUME_SINGLE;
    else { Error("Unable to load plugin \"%s\" (volume integrator)", name.c_str()); *ok = false; vi.kind = RT_VOLUME_EMISSION; }
    ps.ReportUnused();
    return vi;
}
Accelerator MakeAccelerator(const std::string &nameIn, const ParamSet &ps, bool *ok) {
    Accelerator a; *ok = true; std::memset(&a.params, 0, sizeof a.params);
    const std::string &name = nameIn;
    if (void *sym = findPluginSymbol(name, "PbrtHipCreateAccelerator")) {                  // dynload.cpp:246-260 MakeAccelerator
        CParams c{&ps}; PbrtHipAccelerator o{};
        if (reinterpret_cast<PbrtHipCreateAcceleratorFn>(sym)(reinterpret_cast<const PbrtHipParams *>(&c), &kParamsApi, &o) != 0 ||
            (o.params.kind != RT_ACCEL_KDTREE && o.params.kind != RT_ACCEL_GRID)) {
            Error("Unable to load plugin \"%s\" (accelerator)", name.c_str()); *ok = false; a.params.kind = RT_ACCEL_KDTREE;
            a.params.isect_cost = 80; a.params.trav_cost = 1; a.params.empty_bonus = 0.5f; a.params.max_prims = 1; a.params.max_depth = -1;
        } else a.params = o.params;
        ps.ReportUnused();
        return a;
    }
    a.params.kind = RT_ACCEL_KDTREE;
    if (name == "grid") {
        a.params.kind = RT_ACCEL_GRID;
        ps.FindOneBool("refineimmediately", false);
    } else if (name != "kdtree") { Error("Unable to load plugin \"%s\" (accelerator)", name.c_str()); *ok = false; }
    a.params.isect_cost = ps.FindOneInt("intersectcost", 80);              // kdtree.cpp:491-495
    a.params.trav_cost = ps.FindOneInt("traversalcost", 1);
    a.params.empty_bonus = ps.FindOneFloat("emptybonus", 0.5f);
    a.params.max_prims = ps.FindOneInt("maxprims", 1);
    a.params.max_depth = ps.FindOneInt("maxdepth", -1);
    a.params.build_threads = ps.FindOneInt("buildthreads", 0);
    ps.ReportUnused();
    return a;
}

// ------------------------------------------------------------------ camera (perspective.cpp:83-115, camera.cpp:50-70)
bool MakeCamera(const std::string &name, const ParamSet &ps, const Xform &world2cam, const Film &film, RtCamera *out) {
    std::memset(out, 0, sizeof *out);
    if (name == "perspective") out->type = RT_CAMERA_PERSPECTIVE;
    else if (name == "orthographic") out->type = RT_CAMERA_ORTHOGRAPHIC;          // orthographic.cpp:80-112
    else if (name == "environment") out->type = RT_CAMERA_ENVIRONMENT;            // environment.cpp:62-94
    else { Error("Unable to load plugin \"%s\" (camera): \"perspective\", \"orthographic\" and \"environment\" are on the accelerated path", name.c_str()); return false; }
    out->x_res = film.xResolution; out->y_res = film.yResolution;
    float hither = std::fmax(1e-4f, ps.FindOneFloat("hither", 1e-3f));
    float yon = std::fmin(ps.FindOneFloat("yon", 1e30f), 1e30f);
    float shutteropen = ps.FindOneFloat("shutteropen", 0.f);
    float shutterclose = ps.FindOneFloat("shutterclose", 1.f);
    float lensradius = ps.FindOneFloat("lensradius", 0.f);
    float focaldistance = ps.FindOneFloat("focaldistance", 1e30f);
    float frame = ps.FindOneFloat("frameaspectratio", float(film.xResolution) / float(film.yResolution));
    float screen[4];
    if (frame > 1.f) { screen[0] = -frame; screen[1] = frame; screen[2] = -1.f; screen[3] = 1.f; }
    else { screen[0] = -1.f; screen[1] = 1.f; screen[2] = -1.f / frame; screen[3] = 1.f / frame; }
    int swi; const float *sw = ps.FindFloat("screenwindow", &swi);
    if (sw && swi == 4) std::memcpy(screen, sw, 4 * sizeof(float));
    float fov = out->type == RT_CAMERA_PERSPECTIVE ? ps.FindOneFloat("fov", 90.) : 90.f;
    ps.ReportUnused();
    if (out->type == RT_CAMERA_ENVIRONMENT) lensradius = 0.f;                     // "(void) lensradius; // don't need this"
    Xform cameraToScreen = out->type == RT_CAMERA_ORTHOGRAPHIC ? Scale(1.f, 1.f, 1.f / (yon - hither)) * Translate(0.f, 0.f, -hither)   // transform.cpp:177-180
                                                               : Perspective(fov, hither, yon);
    Xform screenToRaster = Scale(float(film.xResolution), float(film.yResolution), 1.f) *
                           Scale(1.f / (screen[1] - screen[0]), 1.f / (screen[2] - screen[3]), 1.f) *
                           Translate(-screen[0], -screen[3], 0.f);
    Xform rasterToScreen = screenToRaster.inverse();
    Xform rasterToCamera = cameraToScreen.inverse() * rasterToScreen;
    Xform cameraToWorld = world2cam.inverse();
    // EnvironmentCamera is not a ProjectiveCamera (environment.cpp:29-38): it has no RasterToCamera, no lens; those fields stay zero
    if (out->type != RT_CAMERA_ENVIRONMENT) {
        std::memcpy(out->raster_to_camera, rasterToCamera.m.m, 16 * sizeof(float));
        out->lens_radius = lensradius; out->focal_distance = focaldistance;
    }
    std::memcpy(out->camera_to_world, cameraToWorld.m.m, 16 * sizeof(float));
    out->hither = hither; out->yon = yon;
    out->shutter_open = shutteropen; out->shutter_close = shutterclose;
    return true;
}

// ------------------------------------------------------------------ API state machine (core/api.cpp)
void SceneDescription::finalize_pointers() {
    scene.n_tris = uint32_t(tri_material.size());
    scene.tri_verts = tri_verts.data(); scene.tri_material = tri_material.data();
    scene.tri_light = tri_light.data(); scene.tri_flags = tri_flags.data();
    scene.n_materials = uint32_t(materials.size()); scene.materials = materials.data();
    scene.n_lights = uint32_t(lights.size()); scene.lights = lights.data();
    scene.n_light_tris = uint32_t(light_tris.size() / 9); scene.light_tris = light_tris.data();
    scene.n_quadrics = uint32_t(quadrics.size()); scene.quadrics = quadrics.empty() ? nullptr : quadrics.data();
    scene.tri_shading = tri_shading.empty() ? nullptr : tri_shading.data();
    scene.n_shading = uint32_t(shading.size()); scene.shading = shading.empty() ? nullptr : shading.data();
    scene.n_xforms = uint32_t(xforms.size() / 32); scene.xforms = xforms.empty() ? nullptr : xforms.data();
}

PbrtApi::PbrtApi() : state(STATE_OPTIONS), nVolumes(0), inObject(false) {
    // RenderOptions defaults api.cpp:62-71.  The reference's default sampler is "bestcandidate" (a
    // precomputed 4096-entry tile pattern, samplers/bestcandidate.cpp) which is out of scope: a scene that does not name
    // one of the supported samplers fails loudly in MakeSampler ("Unable to load plugin") and its frame is invalid.
    filterOpt.name = "mitchell"; filmOpt.name = "image"; samplerOpt.name = "bestcandidate"; accelOpt.name = "kdtree";
    surfOpt.name = "directlighting"; volOpt.name = "emission"; cameraOpt.name = "perspective";
    std::memset(&volume, 0, sizeof volume);
}
PbrtApi::~PbrtApi() { for (SceneDescription *f : frames) delete f; }
void PbrtApi::Diagnostic(int severity, const std::string &msg) { if (severity) Error("%s", msg.c_str()); else Warning("%s", msg.c_str()); }

bool PbrtApi::verifyOptions(const char *fn) {
    if (state == STATE_WORLD) { Error("Options cannot be set inside world block; \"%s\" not allowed.  Ignoring.", fn); return false; }
    return true;
}
bool PbrtApi::verifyWorld(const char *fn) {
    if (state == STATE_OPTIONS) { Error("Scene description must be inside world block; \"%s\" not allowed. Ignoring.", fn); return false; }
    return true;
}
void PbrtApi::Identity() { ctm = Xform(); }
void PbrtApi::Translate(float x, float y, float z) { ctm = ctm * pbrthip::Translate(x, y, z); }
void PbrtApi::Rotate(float a, float x, float y, float z) { ctm = ctm * pbrthip::Rotate(a, x, y, z); }
void PbrtApi::Scale(float x, float y, float z) { ctm = ctm * pbrthip::Scale(x, y, z); }
void PbrtApi::LookAt(const float v[9]) { ctm = ctm * pbrthip::LookAt(v, v + 3, v + 6); }
static Mat4 from_column_major(const float t[16]) {                                  // api.cpp:179-194
    return Mat4(t[0], t[4], t[8], t[12], t[1], t[5], t[9], t[13], t[2], t[6], t[10], t[14], t[3], t[7], t[11], t[15]);
}
void PbrtApi::ConcatTransform(const float m[16]) { ctm = ctm * Xform(from_column_major(m)); }
void PbrtApi::Transform(const float m[16]) { ctm = Xform(from_column_major(m)); }
void PbrtApi::CoordinateSystem(const std::string &n) { named[n] = ctm; }
void PbrtApi::CoordSysTransform(const std::string &n) { if (named.count(n)) ctm = named[n]; }
void PbrtApi::PixelFilter(const std::string &n, const ParamList &p) { if (verifyOptions("PixelFilter")) { filterOpt.name = n; filterOpt.params = ParamSet(p); } }
void PbrtApi::Film(const std::string &n, const ParamList &p) { if (verifyOptions("Film")) { filmOpt.name = n; filmOpt.params = ParamSet(p); } }
void PbrtApi::Sampler(const std::string &n, const ParamList &p) { if (verifyOptions("Sampler")) { samplerOpt.name = n; samplerOpt.params = ParamSet(p); } }
void PbrtApi::Accelerator(const std::string &n, const ParamList &p) { if (verifyOptions("Accelerator")) { accelOpt.name = n; accelOpt.params = ParamSet(p); } }
void PbrtApi::SurfaceIntegrator(const std::string &n, const ParamList &p) { if (verifyOptions("SurfaceIntegrator")) { surfOpt.name = n; surfOpt.params = ParamSet(p); } }
void PbrtApi::VolumeIntegrator(const std::string &n, const ParamList &p) { if (verifyOptions("VolumeIntegrator")) { volOpt.name = n; volOpt.params = ParamSet(p); } }
void PbrtApi::Camera(const std::string &n, const ParamList &p) {
    if (!verifyOptions("Camera")) return;
    cameraOpt.name = n; cameraOpt.params = ParamSet(p);
    worldToCamera = ctm; named["camera"] = ctm.inverse();
}
void PbrtApi::SearchPath(const std::string &path) { if (verifyOptions("SearchPath")) addPluginPath(path); }      // api.cpp:258-262 -> UpdatePluginPath
void PbrtApi::WorldBegin() {
    if (!verifyOptions("WorldBegin")) return;
    state = STATE_WORLD; ctm = Xform(); named["world"] = ctm;
}
void PbrtApi::AttributeBegin() { if (!verifyWorld("AttributeBegin")) return; gsStack.push_back(gs); xfStack.push_back(ctm); }
void PbrtApi::AttributeEnd() {
    if (!verifyWorld("AttributeEnd")) return;
    if (gsStack.empty()) { Error("Unmatched pbrtAttributeEnd() encountered. Ignoring it."); return; }
    gs = gsStack.back(); ctm = xfStack.back(); gsStack.pop_back(); xfStack.pop_back();
}
void PbrtApi::TransformBegin() { if (verifyWorld("TransformBegin")) xfStack.push_back(ctm); }
void PbrtApi::TransformEnd() {
    if (!verifyWorld("TransformEnd")) return;
    if (xfStack.empty()) { Error("Unmatched pbrtTransformEnd() encountered. Ignoring it."); return; }
    ctm = xfStack.back(); xfStack.pop_back();
}
void PbrtApi::Texture(const std::string &name, const std::string &type, const std::string &cls, const ParamList &p) {
    if (!verifyWorld("Texture")) return;
    ParamSet ps(p);
    if (cls != "constant") {                 // textures/*.cpp other than constant.cpp are out of scope
        Error("Unable to load plugin \"%s\" (texture): only \"constant\" textures are on the accelerated path", cls.c_str());
        return;
    }
    if (type == "float") gs.floatTextures[name] = ps.FindOneFloat("value", 1.f);            // textures/constant.cpp
    else if (type == "color") gs.spectrumTextures[name] = ps.FindOneSpectrum("value", Float3{1.f, 1.f, 1.f});
    else Error("Texture type \"%s\" unknown.", type.c_str());
}
void PbrtApi::Material(const std::string &n, const ParamList &p) { if (verifyWorld("Material")) { gs.material = n; gs.materialParams = ParamSet(p); } }
void PbrtApi::AreaLightSource(const std::string &n, const ParamList &p) { if (verifyWorld("AreaLightSource")) { gs.areaLight = n; gs.areaLightParams = ParamSet(p); } }
void PbrtApi::ReverseOrientation() { if (verifyWorld("ReverseOrientation")) gs.reverseOrientation = !gs.reverseOrientation; }

void PbrtApi::LightSource(const std::string &n, const ParamList &p) {
    if (!verifyWorld("LightSource")) return;
    ParamSet ps(p);
    RtLight L; std::memset(&L, 0, sizeof L);
    L.n_samples = 1;
    if (n == "point") {
        // CreateLight lights/point.cpp:78-84 + PointLight ctor :49-54
        Float3 I = ps.FindOneSpectrum("I", Float3{1.f, 1.f, 1.f});
        Float3 P = ps.FindOnePoint("from", Float3{0, 0, 0});
        Xform l2w = pbrthip::Translate(P.x, P.y, P.z) * ctm;
        L.type = RT_LIGHT_POINT; L.color[0] = I.x; L.color[1] = I.y; L.color[2] = I.z;
        const float origin[3] = {0, 0, 0}; l2w.point(origin, L.pos);
    } else if (n == "spot") {
        // CreateLight lights/spot.cpp:95-117 + SpotLight ctor :54-60
        Float3 I = ps.FindOneSpectrum("I", Float3{1.f, 1.f, 1.f});
        float coneangle = ps.FindOneFloat("coneangle", 30.f), conedelta = ps.FindOneFloat("conedeltaangle", 5.f);
        Float3 from = ps.FindOnePoint("from", Float3{0, 0, 0}), to = ps.FindOnePoint("to", Float3{0, 0, 1});
        float dx = to.x - from.x, dy = to.y - from.y, dz = to.z - from.z;
        { float il = 1.f / std::sqrt(dx * dx + dy * dy + dz * dz); dx *= il; dy *= il; dz *= il; }       // Normalize
        float ux, uy, uz;                                                                        // CoordinateSystem geometry.h:324-334
        if (std::fabs(dx) > std::fabs(dy)) { float il = 1.f / std::sqrt(dx * dx + dz * dz); ux = -dz * il; uy = 0.f; uz = dx * il; }
        else { float il = 1.f / std::sqrt(dy * dy + dz * dz); ux = 0.f; uy = dz * il; uz = -dy * il; }
        float vx = (dy * uz) - (dz * uy), vy = (dz * ux) - (dx * uz), vz = (dx * uy) - (dy * ux);
        Xform dirToZ(Mat4(ux, uy, uz, 0, vx, vy, vz, 0, dx, dy, dz, 0, 0, 0, 0, 1));
        Xform l2w = ctm * pbrthip::Translate(from.x, from.y, from.z) * dirToZ.inverse();
        L.type = RT_LIGHT_SPOT; L.color[0] = I.x; L.color[1] = I.y; L.color[2] = I.z;
        const float origin[3] = {0, 0, 0}; l2w.point(origin, L.pos);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L.world_to_light[3 * r + c] = l2w.inv.m[r][c];   // Light::WorldToLight light.h:36
        L.cos_total_width = cosf(pbrthip::radians(coneangle));
        L.cos_falloff_start = cosf(pbrthip::radians(coneangle - conedelta));
    } else if (n == "distant") {
        // CreateLight lights/distant.cpp:94-101 + DistantLight ctor :51-56
        Float3 Lr = ps.FindOneSpectrum("L", Float3{1.f, 1.f, 1.f});
        Float3 from = ps.FindOnePoint("from", Float3{0, 0, 0}), to = ps.FindOnePoint("to", Float3{0, 0, 1});
        const float x = from.x - to.x, y = from.y - to.y, z = from.z - to.z;
        const Mat4 &m = ctm.m;                                                                   // Transform::operator()(Vector) transform.h:94-101
        float wx = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z, wy = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z,
              wz = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z;
        float il = 1.f / std::sqrt(wx * wx + wy * wy + wz * wz);
        L.type = RT_LIGHT_DISTANT; L.color[0] = Lr.x; L.color[1] = Lr.y; L.color[2] = Lr.z;
        L.dir[0] = wx * il; L.dir[1] = wy * il; L.dir[2] = wz * il;
    } else { Error("pbrtLightSource: light type \"%s\" unknown.", n.c_str()); return; }
    ps.ReportUnused();
    lights.push_back(L);
}

Float3 PbrtApi::spectrumParam(const ParamSet &geom, const ParamSet &mat, const std::string &n, Float3 d) {   // paramset.cpp:434-449
    std::string tex = geom.FindTexture(n); if (tex.empty()) tex = mat.FindTexture(n);
    if (!tex.empty()) {
        if (gs.spectrumTextures.count(tex)) return gs.spectrumTextures[tex];
        Error("Couldn't find spectrumtexture named \"%s\"", n.c_str());
    }
    return geom.FindOneSpectrum(n, mat.FindOneSpectrum(n, d));
}
float PbrtApi::floatParam(const ParamSet &geom, const ParamSet &mat, const std::string &n, float d) {         // paramset.cpp:450-465
    std::string tex = geom.FindTexture(n); if (tex.empty()) tex = mat.FindTexture(n);
    if (!tex.empty()) {
        if (gs.floatTextures.count(tex)) return gs.floatTextures[tex];
        Error("Couldn't find float texture named \"%s\"", n.c_str());
    }
    return geom.FindOneFloat(n, mat.FindOneFloat(n, d));
}

int PbrtApi::makeMaterial(const ParamSet &shapeParams) {
    auto clamp0 = [](Float3 c) { Float3 r = {c.x < 0.f ? 0.f : c.x, c.y < 0.f ? 0.f : c.y, c.z < 0.f ? 0.f : c.z}; return r; };
    RtMaterial m; std::memset(&m, 0, sizeof m);
    std::string name = gs.material;
    if (name != "matte" && name != "mirror" && name != "glass" && name != "plastic" && name != "uber") {
        Error("Unable to load plugin \"%s\" (material); using \"matte\" (api.cpp:376-379)", name.c_str());
        name = "matte";
    }
    const ParamSet &mp = gs.materialParams;
    if (floatParam(shapeParams, mp, "bumpmap", 0.f) != 0.f)
        Warning("Non-zero constant \"bumpmap\" displaces uniformly and leaves the shading frame unchanged");
    if (name == "matte") {                                                          // matte.cpp:46-71
        Float3 kd = clamp0(spectrumParam(shapeParams, mp, "Kd", Float3{1.f, 1.f, 1.f}));
        float sig = floatParam(shapeParams, mp, "sigma", 0.f); sig = sig < 0.f ? 0.f : (sig > 90.f ? 90.f : sig);
        m.type = RT_MAT_MATTE; m.kd[0] = kd.x; m.kd[1] = kd.y; m.kd[2] = kd.z; m.sigma = sig; m.ior = 1.f;
    } else if (name == "plastic") {                                                 // plastic.cpp:47-77
        Float3 kd = clamp0(spectrumParam(shapeParams, mp, "Kd", Float3{1.f, 1.f, 1.f}));
        Float3 ks = clamp0(spectrumParam(shapeParams, mp, "Ks", Float3{1.f, 1.f, 1.f}));
        m.type = RT_MAT_PLASTIC; m.kd[0] = kd.x; m.kd[1] = kd.y; m.kd[2] = kd.z; m.ks[0] = ks.x; m.ks[1] = ks.y; m.ks[2] = ks.z;
        m.roughness = floatParam(shapeParams, mp, "roughness", .1f); m.ior = 1.f;
    } else if (name == "uber") {                                                    // uber.cpp:52-100
        Float3 kd = clamp0(spectrumParam(shapeParams, mp, "Kd", Float3{1.f, 1.f, 1.f}));
        Float3 ks = clamp0(spectrumParam(shapeParams, mp, "Ks", Float3{1.f, 1.f, 1.f}));
        Float3 kr = clamp0(spectrumParam(shapeParams, mp, "Kr", Float3{0.f, 0.f, 0.f}));
        Float3 op = clamp0(spectrumParam(shapeParams, mp, "opacity", Float3{1.f, 1.f, 1.f}));
        m.type = RT_MAT_UBER; m.ior = 1.f;
        m.kt[0] = -op.x + 1.f; m.kt[1] = -op.y + 1.f; m.kt[2] = -op.z + 1.f;        // SpecularTransmission(-op + Spectrum(1.), 1., 1.)
        m.kd[0] = op.x * kd.x; m.kd[1] = op.y * kd.y; m.kd[2] = op.z * kd.z;
        m.ks[0] = op.x * ks.x; m.ks[1] = op.y * ks.y; m.ks[2] = op.z * ks.z;
        m.kr[0] = op.x * kr.x; m.kr[1] = op.y * kr.y; m.kr[2] = op.z * kr.z;
        m.roughness = floatParam(shapeParams, mp, "roughness", .1f);
    } else if (name == "mirror") {                                                  // mirror.cpp:42-61
        Float3 kr = clamp0(spectrumParam(shapeParams, mp, "Kr", Float3{1.f, 1.f, 1.f}));
        m.type = RT_MAT_MIRROR; m.kd[0] = kr.x; m.kd[1] = kr.y; m.kd[2] = kr.z; m.ior = 1.f;
    } else {                                                                        // glass.cpp:46-70
        Float3 kr = clamp0(spectrumParam(shapeParams, mp, "Kr", Float3{1.f, 1.f, 1.f}));
        Float3 kt = clamp0(spectrumParam(shapeParams, mp, "Kt", Float3{1.f, 1.f, 1.f}));
        m.type = RT_MAT_GLASS; m.kd[0] = kr.x; m.kd[1] = kr.y; m.kd[2] = kr.z; m.kt[0] = kt.x; m.kt[1] = kt.y; m.kt[2] = kt.z;
        m.ior = floatParam(shapeParams, mp, "index", 1.5f);
    }
    materials.push_back(m);
    return int(materials.size()) - 1;
}

void PbrtApi::Shape(const std::string &n, const ParamList &p) {                     // api.cpp:354-396
    if (!verifyWorld("Shape")) return;
    ParamSet ps(p);
    if (n == "sphere" || n == "disk" || n == "cylinder" || n == "cone" || n == "paraboloid" || n == "hyperboloid") { quadricShape(n, ps); return; }
    if (n != "trianglemesh") {
        Error("Unable to load plugin \"%s\" (shape): \"trianglemesh\" and the quadrics are on the accelerated path (SURVEY.md rows 12-13)", n.c_str());
        return;
    }
    // CreateShape shapes/trianglemesh.cpp:350-406
    int nvi = 0, npi = 0, nuvi = 0;
    const int *vi = ps.FindInt("indices", &nvi);
    const Float3 *P = ps.FindPoint("P", &npi);
    const float *uvs = ps.FindFloat("uv", &nuvi); if (!uvs) uvs = ps.FindFloat("st", &nuvi);
    if (!vi || !P) return;
    // the factory's checks, in its order (trianglemesh.cpp:357-393)
    if (uvs) {
        if (nuvi < 2 * npi) { Error("Not enough of \"uv\"s for triangle mesh.  Expencted %d, found %d.  Discarding.\n", 2 * npi, nuvi); uvs = nullptr; }
        else if (nuvi > 2 * npi) Warning("More \"uv\"s provided than will be used for triangle mesh.  (%d expcted, %d found)\n", 2 * npi, nuvi);
    }
    int nni = 0, nsi = 0;
    const Float3 *S = ps.FindVector("S", &nsi);
    if (S && nsi != npi) { Error("Number of \"S\"s for triangle mesh must match \"P\"s"); S = nullptr; }
    const Float3 *N = ps.FindNormal("N", &nni);
    if (N && nni != npi) { Error("Number of \"N\"s for triangle mesh must match \"P\"s"); N = nullptr; }
    for (int i = 0; i < nvi && (uvs && N); ++i)
        if (vi[i] >= npi || vi[i] < 0) { Error("trianglemesh has out of-bounds vertex index %d (%d \"P\" values were given", vi[i], npi); return; }
    if (uvs && N) {                                                                 // degenerate mappings: discard all uvs
        const int *vp = vi;
        for (int i = 0; i + 2 < nvi; i += 3, vp += 3) {
            const Float3 a = P[vp[0]], b = P[vp[1]], c = P[vp[2]];
            const float e1[3] = {a.x - b.x, a.y - b.y, a.z - b.z}, e2[3] = {c.x - b.x, c.y - b.y, c.z - b.z};
            const float cx = (e1[1] * e2[2]) - (e1[2] * e2[1]), cy = (e1[2] * e2[0]) - (e1[0] * e2[2]), cz = (e1[0] * e2[1]) - (e1[1] * e2[0]);
            const float area = .5f * sqrtf(cx * cx + cy * cy + cz * cz);
            if (area < 1e-7) continue;
            if ((uvs[2 * vp[0]] == uvs[2 * vp[1]] && uvs[2 * vp[0] + 1] == uvs[2 * vp[1] + 1]) ||
                (uvs[2 * vp[1]] == uvs[2 * vp[2]] && uvs[2 * vp[1] + 1] == uvs[2 * vp[2] + 1]) ||
                (uvs[2 * vp[2]] == uvs[2 * vp[0]] && uvs[2 * vp[2] + 1] == uvs[2 * vp[0] + 1])) {
                Warning("Degenerate uv coordinates in triangle mesh.  Discarding all uvs.");
                uvs = nullptr;
                break;
            }
        }
    }
    for (int i = 0; i < nvi; ++i)
        if (vi[i] >= npi || vi[i] < 0) { Error("trianglemesh has out of-bounds vertex index %d (%d \"P\" values were given", vi[i], npi); return; }
    if (inObject) { Error("Object instancing is not on the accelerated path (SURVEY.md row 9); shape ignored"); return; }
    ps.ReportUnused();
    const int ntris = nvi / 3;
    std::vector<float> world(size_t(npi) * 3);
    for (int i = 0; i < npi; ++i) ctm.point(&P[i].x, &world[size_t(3) * i]);        // trianglemesh.cpp:166-168
    Mesh mesh;
    mesh.flags = uint8_t((gs.reverseOrientation ^ ctm.swaps_handedness()) ? 1 : 0);   // shape.cpp:27-35,49-50
    mesh.verts.resize(size_t(ntris) * 9);
    for (int t = 0; t < ntris; ++t)
        for (int k = 0; k < 3; ++k) std::memcpy(&mesh.verts[size_t(t) * 9 + 3 * k], &world[size_t(3) * vi[3 * t + k]], 3 * sizeof(float));
    if (uvs || N || S) {                                                            // TriangleMesh ctor :141-169 keeps uv / N / S per vertex
        mesh.shading.resize(size_t(ntris));
        std::memcpy(mesh.o2w, ctm.m.m, 16 * sizeof(float)); std::memcpy(mesh.o2w + 16, ctm.inv.m, 16 * sizeof(float));
        for (int t = 0; t < ntris; ++t) {
            RtTriShading &r = mesh.shading[size_t(t)]; std::memset(&r, 0, sizeof r);
            r.flags = (uvs ? RT_SHADING_UV : 0) | (N ? RT_SHADING_N : 0) | (S ? RT_SHADING_S : 0);
            const float def[6] = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f};                    // GetUVs :321-326
            for (int k = 0; k < 3; ++k) {
                const int v = vi[3 * t + k];
                r.uv[2 * k] = uvs ? uvs[2 * v] : def[2 * k]; r.uv[2 * k + 1] = uvs ? uvs[2 * v + 1] : def[2 * k + 1];
                if (N) { r.n[3 * k] = N[v].x; r.n[3 * k + 1] = N[v].y; r.n[3 * k + 2] = N[v].z; }
                if (S) { r.s[3 * k] = S[v].x; r.s[3 * k + 1] = S[v].y; r.s[3 * k + 2] = S[v].z; }
            }
        }
    }
    mesh.light = -1;
    if (!gs.areaLight.empty()) {                                                    // api.cpp:362-366, area.cpp:106-111
        if (gs.areaLight != "area") Error("Unable to load plugin \"%s\" (area light)", gs.areaLight.c_str());
        else if (ntris > 0) {
            Float3 Le = gs.areaLightParams.FindOneSpectrum("L", Float3{1.f, 1.f, 1.f});
            int ns = gs.areaLightParams.FindOneInt("nsamples", 1);
            RtLight L; std::memset(&L, 0, sizeof L);
            L.type = RT_LIGHT_AREA; L.color[0] = Le.x; L.color[1] = Le.y; L.color[2] = Le.z; L.n_samples = std::max(1, ns);
            L.first_tri = uint32_t(light_tris.size() / 9); L.n_tris = uint32_t(ntris);
            L.reverse_orientation = gs.reverseOrientation ? 1 : 0; L.flip_normal = mesh.flags;
            // AreaLight ctor area.cpp:33-54: the refinement stack pops the LAST triangle first
            for (int t = ntris - 1; t >= 0; --t) light_tris.insert(light_tris.end(), &mesh.verts[size_t(t) * 9], &mesh.verts[size_t(t) * 9] + 9);
            mesh.light = int(lights.size());
            lights.push_back(L);
        }
    }
    mesh.material = makeMaterial(ps);
    meshes.push_back(std::move(mesh));
}

// CreateShape + ctor of shapes/sphere.cpp:255-264,:89-99, disk.cpp:131-138,:51-59, cylinder.cpp:183-190,:52-59 and
// Shape::WorldBound (shape.h:57-59, transform.cpp:148-159)
void PbrtApi::quadricShape(const std::string &name, const ParamSet &ps) {
    if (inObject) { Error("Object instancing is not on the accelerated path (SURVEY.md row 9); shape ignored"); return; }
    auto clampf = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
    RtQuadric q; std::memset(&q, 0, sizeof q);
    std::memcpy(q.object_to_world, ctm.m.m, sizeof q.object_to_world);
    std::memcpy(q.world_to_object, ctm.inv.m, sizeof q.world_to_object);
    float lo[3], hi[3];
    if (name == "sphere") {
        const float radius = ps.FindOneFloat("radius", 1.f);
        const float z0 = ps.FindOneFloat("zmin", -radius), z1 = ps.FindOneFloat("zmax", radius);
        const float pm = ps.FindOneFloat("phimax", 360.f);
        q.type = RT_QUADRIC_SPHERE;
        q.radius = radius;
        q.zmin = clampf(std::fmin(z0, z1), -radius, radius);
        q.zmax = clampf(std::fmax(z0, z1), -radius, radius);
        q.theta_min = acosf(clampf(q.zmin / radius, -1.f, 1.f));
        q.theta_max = acosf(clampf(q.zmax / radius, -1.f, 1.f));
        q.phi_max = pbrthip::radians(clampf(pm, 0.0f, 360.0f));
        lo[0] = -radius; lo[1] = -radius; lo[2] = q.zmin; hi[0] = radius; hi[1] = radius; hi[2] = q.zmax;
    } else if (name == "disk") {
        const float height = ps.FindOneFloat("height", 0.f), radius = ps.FindOneFloat("radius", 1.f);
        const float inner = ps.FindOneFloat("innerradius", 0.f), pm = ps.FindOneFloat("phimax", 360.f);
        q.type = RT_QUADRIC_DISK;
        q.radius = radius; q.zmin = height; q.zmax = inner;
        q.phi_max = pbrthip::radians(clampf(pm, 0.0f, 360.0f));
        lo[0] = -radius; lo[1] = -radius; lo[2] = height; hi[0] = radius; hi[1] = radius; hi[2] = height;
    } else if (name == "cone") {                                                    // cone.cpp:187-194, :41-48
        const float radius = ps.FindOneFloat("radius", 1.f), height = ps.FindOneFloat("height", 1.f), pm = ps.FindOneFloat("phimax", 360.f);
        q.type = RT_QUADRIC_CONE;
        q.radius = radius; q.zmin = 0.f; q.zmax = height;
        q.phi_max = pbrthip::radians(clampf(pm, 0.0f, 360.0f));
        lo[0] = -radius; lo[1] = -radius; lo[2] = 0.f; hi[0] = radius; hi[1] = radius; hi[2] = height;
    } else if (name == "paraboloid") {                                              // paraboloid.cpp:191-199, :44-52
        const float radius = ps.FindOneFloat("radius", 1.f);
        const float z0 = ps.FindOneFloat("zmin", 0.f), z1 = ps.FindOneFloat("zmax", 1.f), pm = ps.FindOneFloat("phimax", 360.f);
        q.type = RT_QUADRIC_PARABOLOID;
        q.radius = radius; q.zmin = std::fmin(z0, z1); q.zmax = std::fmax(z0, z1);
        q.phi_max = pbrthip::radians(clampf(pm, 0.0f, 360.0f));
        lo[0] = -radius; lo[1] = -radius; lo[2] = q.zmin; hi[0] = radius; hi[1] = radius; hi[2] = q.zmax;
    } else if (name == "hyperboloid") {                                             // hyperboloid.cpp:240-246, :46-70
        Float3 p1 = ps.FindOnePoint("p1", Float3{0, 0, 0}), p2 = ps.FindOnePoint("p2", Float3{1, 1, 1});
        const float pm = ps.FindOneFloat("phimax", 360.f);
        q.type = RT_QUADRIC_HYPERBOLOID;
        q.phi_max = pbrthip::radians(clampf(pm, 0.0f, 360.0f));
        const float rad1 = std::sqrt(p1.x * p1.x + p1.y * p1.y), rad2 = std::sqrt(p2.x * p2.x + p2.y * p2.y);
        q.radius = std::fmax(rad1, rad2);
        q.zmin = std::fmin(p1.z, p2.z); q.zmax = std::fmax(p1.z, p2.z);
        if (p2.z == 0.) std::swap(p1, p2);
        Float3 pp = p1; float xy1, xy2, a, c;
        do {
            pp.x += 2.f * (p2.x - p1.x); pp.y += 2.f * (p2.y - p1.y); pp.z += 2.f * (p2.z - p1.z);
            xy1 = pp.x * pp.x + pp.y * pp.y;
            xy2 = p2.x * p2.x + p2.y * p2.y;
            a = (1.f / xy1 - (pp.z * pp.z) / (xy1 * p2.z * p2.z)) / (1 - (xy2 * pp.z * pp.z) / (xy1 * p2.z * p2.z));
            c = (a * xy2 - 1) / (p2.z * p2.z);
        } while (std::isinf(a) || std::isnan(a));
        q.p1[0] = p1.x; q.p1[1] = p1.y; q.p1[2] = p1.z; q.p2[0] = p2.x; q.p2[1] = p2.y; q.p2[2] = p2.z; q.a = a; q.c = c;
        lo[0] = -q.radius; lo[1] = -q.radius; lo[2] = q.zmin; hi[0] = q.radius; hi[1] = q.radius; hi[2] = q.zmax;
    } else {
        const float radius = ps.FindOneFloat("radius", 1.f);
        const float z0 = ps.FindOneFloat("zmin", -1.f), z1 = ps.FindOneFloat("zmax", 1.f), pm = ps.FindOneFloat("phimax", 360.f);
        q.type = RT_QUADRIC_CYLINDER;
        q.radius = radius; q.zmin = std::fmin(z0, z1); q.zmax = std::fmax(z0, z1);
        q.phi_max = pbrthip::radians(clampf(pm, 0.0f, 360.0f));
        lo[0] = -radius; lo[1] = -radius; lo[2] = q.zmin; hi[0] = radius; hi[1] = radius; hi[2] = q.zmax;
    }
    ps.ReportUnused();
    // world bound of the object bound BBox(lo, hi)
    float bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int c = 0; c < 8; ++c) {
        const float pt[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
        float w[3]; ctm.point(pt, w);
        for (int a = 0; a < 3; ++a) { bmin[a] = std::fmin(bmin[a], w[a]); bmax[a] = std::fmax(bmax[a], w[a]); }
    }
    Mesh mesh;
    mesh.flags = uint8_t(((gs.reverseOrientation ^ ctm.swaps_handedness()) ? 1 : 0) | 2);
    mesh.verts = {bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2], bmin[0], bmin[1], bmin[2]};
    mesh.light = -1;
    if (!gs.areaLight.empty() && q.type > RT_QUADRIC_CYLINDER)
        Error("Area lights on cone / paraboloid / hyperboloid: the reference's Shape::Sample is unimplemented for them (shape.h:84-88); rendered as a non-emitting surface");
    else if (!gs.areaLight.empty()) {                                               // api.cpp:362-366: AreaLight keeps a CanIntersect shape as is (area.cpp:38-39)
        if (gs.areaLight != "area") Error("Unable to load plugin \"%s\" (area light)", gs.areaLight.c_str());
        else {
            Float3 Le = gs.areaLightParams.FindOneSpectrum("L", Float3{1.f, 1.f, 1.f});
            int ns = gs.areaLightParams.FindOneInt("nsamples", 1);
            RtLight L; std::memset(&L, 0, sizeof L);
            L.type = RT_LIGHT_AREA; L.color[0] = Le.x; L.color[1] = Le.y; L.color[2] = Le.z; L.n_samples = std::max(1, ns);
            L.first_tri = 0; L.n_tris = 0; L.quadric_plus1 = int32_t(quadrics.size()) + 1;
            L.reverse_orientation = gs.reverseOrientation ? 1 : 0; L.flip_normal = mesh.flags & 1;
            mesh.light = int(lights.size());
            lights.push_back(L);
        }
    }
    mesh.material = makeMaterial(ps);
    quadrics.push_back(q);
    meshes.push_back(std::move(mesh));
}

void PbrtApi::Volume(const std::string &n, const ParamList &p) {                    // api.cpp:403-409, homogeneous.cpp:76-88
    if (!verifyWorld("Volume")) return;
    ParamSet ps(p);
    if (n != "homogeneous") { Error("Unable to load plugin \"%s\" (volume region): only \"homogeneous\" is on the accelerated path", n.c_str()); return; }
    if (nVolumes++ > 0) { Error("Only one volume region is supported on the accelerated path (AggregateVolume is out of scope)"); return; }
    Float3 sa = ps.FindOneSpectrum("sigma_a", Float3{0, 0, 0}), ss = ps.FindOneSpectrum("sigma_s", Float3{0, 0, 0});
    float g = ps.FindOneFloat("g", 0.);
    Float3 Le = ps.FindOneSpectrum("Le", Float3{0, 0, 0});
    Float3 p0 = ps.FindOnePoint("p0", Float3{0, 0, 0}), p1 = ps.FindOnePoint("p1", Float3{1, 1, 1});
    ps.ReportUnused();
    volume.present = 1;
    Xform w2v = ctm.inverse();
    std::memcpy(volume.world_to_volume, w2v.m.m, 16 * sizeof(float));
    // BBox(p0,p1) geometry.h:237-244
    volume.p0[0] = std::fmin(p0.x, p1.x); volume.p0[1] = std::fmin(p0.y, p1.y); volume.p0[2] = std::fmin(p0.z, p1.z);
    volume.p1[0] = std::fmax(p0.x, p1.x); volume.p1[1] = std::fmax(p0.y, p1.y); volume.p1[2] = std::fmax(p0.z, p1.z);
    volume.sigma_a[0] = sa.x; volume.sigma_a[1] = sa.y; volume.sigma_a[2] = sa.z;
    volume.sigma_s[0] = ss.x; volume.sigma_s[1] = ss.y; volume.sigma_s[2] = ss.z;
    volume.le[0] = Le.x; volume.le[1] = Le.y; volume.le[2] = Le.z; volume.g = g;
}
void PbrtApi::ObjectBegin(const std::string &) { if (!verifyWorld("ObjectBegin")) return; AttributeBegin(); inObject = true; }
void PbrtApi::ObjectEnd() { if (!verifyWorld("ObjectEnd")) return; inObject = false; AttributeEnd(); }
void PbrtApi::ObjectInstance(const std::string &n) { if (verifyWorld("ObjectInstance")) Error("Object instancing is not on the accelerated path; instance \"%s\" ignored", n.c_str()); }

void PbrtApi::resetWorld() {
    meshes.clear(); materials.clear(); lights.clear(); light_tris.clear(); quadrics.clear();
    std::memset(&volume, 0, sizeof volume); nVolumes = 0;
}

void PbrtApi::WorldEnd() {                                                          // api.cpp:458-529
    if (!verifyWorld("WorldEnd")) return;
    while (!gsStack.empty()) { Warning("Missing end to pbrtAttributeBegin()"); gsStack.pop_back(); xfStack.pop_back(); }
    SceneDescription *sd = new SceneDescription();
    bool ok, all = true;
    Filter filter = MakeFilter(filterOpt.name, filterOpt.params, &ok); all &= ok;
    sd->film = MakeFilm(filmOpt.name, filmOpt.params, filter, &ok); all &= ok;
    std::memset(&sd->scene, 0, sizeof sd->scene); std::memset(&sd->render, 0, sizeof sd->render);
    all &= MakeCamera(cameraOpt.name, cameraOpt.params, worldToCamera, sd->film, &sd->scene.camera);
    pbrthip::Sampler smp = MakeSampler(samplerOpt.name, samplerOpt.params, sd->film, &ok); all &= ok;
    pbrthip::SurfaceIntegrator si = MakeSurfaceIntegrator(surfOpt.name, surfOpt.params, &ok); all &= ok;
    pbrthip::VolumeIntegrator vi = MakeVolumeIntegrator(volOpt.name, volOpt.params, &ok); all &= ok;
    pbrthip::Accelerator acc = MakeAccelerator(accelOpt.name, accelOpt.params, &ok);
    if (!ok) { ParamSet none; acc = MakeAccelerator("kdtree", none, &ok); }       // api.cpp:497-502
    if (!all) Error("Unable to create scene due to missing plug-ins");
    if (lights.empty()) Warning("No light sources defined in scene; possibly rendering a black image.");

    // primitives in KdTreeAccel order: per mesh, last triangle first (primitive.cpp:40-53, kdtree.cpp:146-148)
    bool any_shading = false;
    for (const Mesh &m : meshes) any_shading = any_shading || !m.shading.empty();
    for (const Mesh &m : meshes) {
        const int nt = int(m.verts.size() / 9);
        uint32_t xf = 0;
        if (!m.shading.empty()) { xf = uint32_t(sd->xforms.size() / 32); sd->xforms.insert(sd->xforms.end(), m.o2w, m.o2w + 32); }
        for (int t = nt - 1; t >= 0; --t) {
            sd->tri_verts.insert(sd->tri_verts.end(), &m.verts[size_t(t) * 9], &m.verts[size_t(t) * 9] + 9);
            sd->tri_material.push_back(uint16_t(m.material)); sd->tri_light.push_back(m.light); sd->tri_flags.push_back(m.flags);
            if (any_shading) {
                if (m.shading.empty()) sd->tri_shading.push_back(-1);
                else { sd->tri_shading.push_back(int32_t(sd->shading.size())); sd->shading.push_back(m.shading[size_t(t)]); sd->shading.back().xform = xf; }
            }
        }
    }
    if (materials.size() > 65535) { Error("more than 65535 material instances"); all = false; }
    sd->materials = materials; sd->lights = lights; sd->light_tris = light_tris; sd->quadrics = quadrics;
    sd->scene.volume = volume; sd->scene.accel = acc.params;
    RtRenderDesc &r = sd->render;
    r.integrator = si.kind; r.max_depth = si.maxDepth; r.strategy = si.strategy;
    r.volume_integrator = vi.kind; r.step_size = vi.stepSize;
    r.sampler = smp.kind; r.x_samples = smp.xsamples; r.y_samples = smp.ysamples; r.jitter = smp.jitter; r.pixel_samples = smp.pixelsamples;
    r.seed = smp.seed;
    r.x_res = sd->film.xResolution; r.y_res = sd->film.yResolution;
    r.x_pixel_start = sd->film.xPixelStart; r.y_pixel_start = sd->film.yPixelStart;
    r.x_pixel_count = sd->film.xPixelCount; r.y_pixel_count = sd->film.yPixelCount;
    sd->film.GetSampleExtent(&r.x_start, &r.x_end, &r.y_start, &r.y_end);
    r.filter_x_width = filter.xWidth; r.filter_y_width = filter.yWidth;
    std::memcpy(r.filter_table, sd->film.filterTable, sizeof r.filter_table);
    r.shard_index = 0; r.shard_count = 1; r.tile_pixels = 64;
    sd->finalize_pointers();
    sd->valid = all;
    frames.push_back(sd);
    // api.cpp:476-482
    state = STATE_OPTIONS; ctm = Xform(); named.clear(); resetWorld();
}

}  // namespace pbrthip

// ------------------------------------------------------------------ C entry points for the Python harness
using namespace pbrthip;
// the library is compiled -fvisibility=hidden: these C entry points are its whole dynamic surface
#pragma GCC visibility push(default)
extern "C" {
struct PbrtHostScene { PbrtApi api; };

PbrtHostScene *pbrt_host_parse_file(const char *path, int quiet) {
    SetQuiet(quiet != 0); ResetDiagnostics();
    PbrtHostScene *h = new PbrtHostScene();
    SceneParser parser(h->api);
    parser.ParseFile(path);
    return h;
}
PbrtHostScene *pbrt_host_parse_string(const char *text, int quiet) {
    SetQuiet(quiet != 0); ResetDiagnostics();
    PbrtHostScene *h = new PbrtHostScene();
    SceneParser parser(h->api);
    parser.ParseString(text);
    return h;
}
void pbrt_host_free(PbrtHostScene *h) { delete h; }
int pbrt_host_frame_count(PbrtHostScene *h) { return int(h->api.frames.size()); }
int pbrt_host_frame_valid(PbrtHostScene *h, int i) { return h->api.frames[i]->valid ? 1 : 0; }
const RtSceneDesc *pbrt_host_scene_desc(PbrtHostScene *h, int i) { return &h->api.frames[i]->scene; }
RtRenderDesc *pbrt_host_render_desc(PbrtHostScene *h, int i) { return &h->api.frames[i]->render; }
int pbrt_host_premultiply(PbrtHostScene *h, int i) { return h->api.frames[i]->film.premultiplyAlpha ? 1 : 0; }
const char *pbrt_host_filename(PbrtHostScene *h, int i) { return h->api.frames[i]->film.filename.c_str(); }
int pbrt_host_warnings() { return WarningCount(); }
int pbrt_host_errors() { return ErrorCount(); }
// field access without mirroring struct layouts in Python
void pbrt_host_set_shard(RtRenderDesc *r, int index, int count, int tile_pixels) { r->shard_index = index; r->shard_count = count; r->tile_pixels = tile_pixels; }
void pbrt_host_set_seed(RtRenderDesc *r, unsigned seed) { r->seed = seed; }
void pbrt_host_film_dims(const RtRenderDesc *r, int *out8) {
    out8[0] = r->x_pixel_count; out8[1] = r->y_pixel_count; out8[2] = r->x_start; out8[3] = r->x_end;
    out8[4] = r->y_start; out8[5] = r->y_end; out8[7] = r->integrator;
    if (r->sampler == RT_SAMPLER_LOWDISCREPANCY) {          // LDSampler rounds up to a power of two (lowdiscrepancy.cpp:57-66)
        unsigned v = unsigned(r->pixel_samples); v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; out8[6] = int(v + 1);
    } else out8[6] = r->x_samples * r->y_samples;
}
void pbrt_host_scene_counts(const RtSceneDesc *s, unsigned *out4) { out4[0] = s->n_tris; out4[1] = s->n_materials; out4[2] = s->n_lights; out4[3] = s->n_light_tris; }
const float *pbrt_host_camera(const RtSceneDesc *s) { return s->camera.raster_to_camera; }
const float *pbrt_host_tri_verts(const RtSceneDesc *s) { return s->tri_verts; }
const RtAccelParams *pbrt_host_accel_params(const RtSceneDesc *s) { return &s->accel; }
// canonical byte image of a frame's descriptors (include/pbrt_hip_desc.h); returns the size, writes when the buffer is large enough
long long pbrt_host_serialize(PbrtHostScene *h, int i, unsigned char *out, long long cap) {
    const size_t n = rt_desc_serialize(&h->api.frames[i]->scene, &h->api.frames[i]->render, nullptr);
    if (out && (long long)n <= cap) rt_desc_serialize(&h->api.frames[i]->scene, &h->api.frames[i]->render, out);
    return (long long)n;
}
int pbrt_host_shading(const RtSceneDesc *s, const int32_t **idx, const RtTriShading **rec, const float **xforms, unsigned *n_rec, unsigned *n_xf) {
    *idx = s->tri_shading; *rec = s->shading; *xforms = s->xforms; *n_rec = s->n_shading; *n_xf = s->n_xforms; return 0;
}
// scene-text writers for the synthetic generators (pbrt-v1_amd/scenes.py): "%.9g" round-trips every float32; np.savetxt
// needs 24 s for the 9 M numbers of a 1 M-triangle mesh, snprintf 1 s.  Returns the bytes written (excluding the NUL), or -1.
long long pbrt_host_format_f32(const float *v, long long n, int per_line, char *out, long long cap) {
    long long w = 0;
    for (long long i = 0; i < n; ++i) {
        if (cap - w < 32) return -1;
        w += std::snprintf(out + w, size_t(cap - w), "%.9g", double(v[i]));
        out[w++] = ((i + 1) % per_line == 0) ? '\n' : ' ';
    }
    if (w < cap) out[w] = 0;
    return w;
}
long long pbrt_host_format_iota(long long first, long long n, int per_line, char *out, long long cap) {
    long long w = 0;
    for (long long i = 0; i < n; ++i) {
        if (cap - w < 32) return -1;
        w += std::snprintf(out + w, size_t(cap - w), "%lld", first + i);
        out[w++] = ((i + 1) % per_line == 0) ? '\n' : ' ';
    }
    if (w < cap) out[w] = 0;
    return w;
}
// Film output: the reference's WriteRGBAImage (core/exrio.cpp:75-96) and the tile merge of tools/exrassemble.cpp
int pbrt_host_write_exr(const char *path, const float *rgb, const float *alpha, int xRes, int yRes, int totalX, int totalY, int xOff, int yOff) {
    return WriteRGBAImage(path, rgb, alpha, xRes, yRes, totalX, totalY, xOff, yOff) ? 0 : -1;
}
int pbrt_host_read_exr_info(const char *path, int *out6) {
    ExrImage img; if (!ReadRGBAImage(path, img)) return -1;
    out6[0] = img.xRes; out6[1] = img.yRes; out6[2] = img.totalXRes; out6[3] = img.totalYRes; out6[4] = img.xOffset; out6[5] = img.yOffset; return 0;
}
// tools/exrassemble.cpp: merge crop-window renders (one EXR per tile / process) into the full image; `paths` is newline-separated
float pbrt_host_assemble_exr(const char *paths, const char *out) {
    std::vector<std::string> in; std::string cur;
    for (const char *c = paths; ; ++c) { if (*c == '\n' || *c == 0) { if (!cur.empty()) in.push_back(cur); cur.clear(); if (!*c) break; } else cur.push_back(*c); }
    return AssembleRGBAImages(in, out);
}
int pbrt_host_read_exr(const char *path, float *rgb, float *alpha) {
    ExrImage img; if (!ReadRGBAImage(path, img)) return -1;
    std::memcpy(rgb, img.rgb.data(), img.rgb.size() * sizeof(float)); std::memcpy(alpha, img.alpha.data(), img.alpha.size() * sizeof(float)); return 0;
}
}
#pragma GCC visibility pop
