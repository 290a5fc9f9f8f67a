// scene_api.h -- host-side mirror of pbrt-v1's scene-description API and plugin factories for the
// hot path, ending in the flat RtSceneDesc / RtRenderDesc the HIP library consumes.
//
// Reference interface mirrored (names, argument meaning, defaults, diagnostics):
//   * the RenderMan-style state machine of core/api.cpp:109-483 (options block / world block, CTM and
//     graphics-state stacks, named coordinate systems) -- one method per pbrt* function of core/api.h:29-85;
//   * the Make* plugin factories of core/dynload.cpp:112-260 for the plugins on the hot path
//     (SURVEY.md section 8a): filters box/gaussian/mitchell/sinc/triangle, film "image", camera
//     "perspective", samplers stratified/lowdiscrepancy/random, surface integrators
//     whitted/directlighting/path, volume integrators emission/single, accelerators kdtree/grid, shape
//     "trianglemesh", materials matte/glass/mirror, lights point + area, volume "homogeneous";
//     each takes the same parameters with the same defaults as the reference factory it replaces.
//   * RenderOptions::MakeScene (core/api.cpp:484-529): instead of building C++ objects it flattens.
// Plugins the reference has but this path does not accelerate produce the reference's own
// "unknown / missing plug-in" style Error and the documented fallback (matte / kdtree), never a silent
// substitute.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "../../../include/pbrt_hip.h"
#include "host_math.h"
#include "paramset.h"

namespace pbrthip {

// ---- plugin descriptors (what the Create* factories of the reference would have returned) ----
struct Filter {               // core/sampling.h:102-115
    std::string name; float xWidth, yWidth, invXWidth, invYWidth;
    float p0 = 0, p1 = 0, p2 = 0, p3 = 0;   // per-filter parameters
    float Evaluate(float x, float y) const;
};
struct Film {                 // film/image.cpp:69-101,213-233
    int xResolution, yResolution; float cropWindow[4]; std::string filename; bool premultiplyAlpha; int writeFrequency;
    int xPixelStart, yPixelStart, xPixelCount, yPixelCount;
    Filter filter; float filterTable[256];
    void GetSampleExtent(int *xs, int *xe, int *ys, int *ye) const;
};
struct Sampler { int kind; int xsamples, ysamples; bool jitter; int pixelsamples; unsigned seed; };
struct SurfaceIntegrator { int kind; int maxDepth; int strategy; };
struct VolumeIntegrator { int kind; float stepSize; };
struct Accelerator { RtAccelParams params; };

struct SceneDescription {
    // flattened arrays referenced by `scene`
    std::vector<float> tri_verts; std::vector<uint16_t> tri_material; std::vector<int32_t> tri_light; std::vector<uint8_t> tri_flags;
    std::vector<RtMaterial> materials; std::vector<RtLight> lights; std::vector<float> light_tris; std::vector<RtQuadric> quadrics;
    std::vector<int32_t> tri_shading; std::vector<RtTriShading> shading; std::vector<float> xforms;   // per-vertex uv / N / S (trianglemesh)
    RtSceneDesc scene; RtRenderDesc render;
    Film film;
    bool valid = false;
    void finalize_pointers();
};

Filter MakeFilter(const std::string &name, const ParamSet &ps, bool *ok);
Film MakeFilm(const std::string &name, const ParamSet &ps, const Filter &f, bool *ok);
Sampler MakeSampler(const std::string &name, const ParamSet &ps, const Film &film, bool *ok);
SurfaceIntegrator MakeSurfaceIntegrator(const std::string &name, const ParamSet &ps, bool *ok);
VolumeIntegrator MakeVolumeIntegrator(const std::string &name, const ParamSet &ps, bool *ok);
Accelerator MakeAccelerator(const std::string &name, const ParamSet &ps, bool *ok);
bool MakeCamera(const std::string &name, const ParamSet &ps, const Xform &world2cam, const Film &film, RtCamera *out);

class PbrtApi : public DirectiveSink {
  public:
    PbrtApi();
    // DirectiveSink == the pbrt* API (core/api.h:29-85)
    void Identity() override;
    void Translate(float x, float y, float z) override;
    void Rotate(float a, float x, float y, float z) override;
    void Scale(float x, float y, float z) override;
    void LookAt(const float v[9]) override;
    void ConcatTransform(const float m[16]) override;
    void Transform(const float m[16]) override;
    void CoordinateSystem(const std::string &n) override;
    void CoordSysTransform(const std::string &n) override;
    void PixelFilter(const std::string &n, const ParamList &p) override;
    void Film(const std::string &n, const ParamList &p) override;
    void Sampler(const std::string &n, const ParamList &p) override;
    void Accelerator(const std::string &n, const ParamList &p) override;
    void SurfaceIntegrator(const std::string &n, const ParamList &p) override;
    void VolumeIntegrator(const std::string &n, const ParamList &p) override;
    void Camera(const std::string &n, const ParamList &p) override;
    void SearchPath(const std::string &n) override;
    void WorldBegin() override;
    void AttributeBegin() override;
    void AttributeEnd() override;
    void TransformBegin() override;
    void TransformEnd() override;
    void Texture(const std::string &name, const std::string &type, const std::string &cls, const ParamList &p) override;
    void Material(const std::string &n, const ParamList &p) override;
    void LightSource(const std::string &n, const ParamList &p) override;
    void AreaLightSource(const std::string &n, const ParamList &p) override;
    void Shape(const std::string &n, const ParamList &p) override;
    void ReverseOrientation() override;
    void Volume(const std::string &n, const ParamList &p) override;
    void ObjectBegin(const std::string &n) override;
    void ObjectEnd() override;
    void ObjectInstance(const std::string &n) override;
    void WorldEnd() override;
    void Diagnostic(int severity, const std::string &msg) override;

    // the frames described so far (one per WorldEnd)
    std::vector<SceneDescription *> frames;
    ~PbrtApi();

  private:
    enum { STATE_OPTIONS = 1, STATE_WORLD = 2 };
    int state;
    Xform ctm;
    std::map<std::string, Xform> named;
    struct Named { std::string name; ParamSet params; };
    Named filterOpt, filmOpt, samplerOpt, accelOpt, surfOpt, volOpt, cameraOpt;
    Xform worldToCamera;
    struct GraphicsState {
        std::map<std::string, float> floatTextures;            // constant textures only (texture.h:113-123)
        std::map<std::string, Float3> spectrumTextures;
        ParamSet materialParams; std::string material = "matte";
        ParamSet areaLightParams; std::string areaLight;
        bool reverseOrientation = false;
    } gs;
    std::vector<GraphicsState> gsStack; std::vector<Xform> xfStack;
    // world being accumulated
    struct Mesh { std::vector<float> verts; int material; int light; uint8_t flags;
                  std::vector<RtTriShading> shading; float o2w[32]; };                  // shading: empty or one record per triangle   // 9 floats per triangle, API order;
                                                                                        // flags bit1: a quadric (one slot = its world bound)
    std::vector<RtQuadric> quadrics;
    std::vector<Mesh> meshes;
    std::vector<RtMaterial> materials; std::vector<RtLight> lights; std::vector<float> light_tris;
    RtVolume volume; int nVolumes;
    bool inObject;
    bool verifyOptions(const char *fn); bool verifyWorld(const char *fn);
    int makeMaterial(const ParamSet &shapeParams);
    void quadricShape(const std::string &name, const ParamSet &ps);
    Float3 spectrumParam(const ParamSet &geom, const ParamSet &mat, const std::string &n, Float3 d);
    float floatParam(const ParamSet &geom, const ParamSet &mat, const std::string &n, float d);
    void resetWorld();
};

// diagnostics counters (tests assert on them)
int WarningCount(); int ErrorCount(); void ResetDiagnostics(); void SetQuiet(bool q);

}  // namespace pbrthip
