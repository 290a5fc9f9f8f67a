/* pbrt_hip_plugin.h -- the product-side plugin ABI of the host library (libpbrt_host.so), plain C.
 *
 * pbrt-v1 loads every SurfaceIntegrator / VolumeIntegrator / Sampler / Accelerator ("Aggregate") from a shared object named after the
 * plugin, found along the search path (PBRT_SEARCHPATH, pbrtSearchPath), through an extern "C" factory `Create<Kind>`:
 *     core/dynload.cpp:41-61    the Create* function-pointer types            core/dynload.cpp:462-514  the per-kind loaders (GetPlugin<>)
 *     core/dynload.cpp:185-260  MakeSurfaceIntegrator / MakeVolumeIntegrator / MakeSampler / MakeAccelerator
 * The host library keeps that mechanism for the plugin kinds of the hot path: `SurfaceIntegrator "name"` first looks for <name>.so (or
 * lib<name>.so) in the directories of PBRT_HIP_PLUGIN_PATH and of the scene's SearchPath directives, resolves the factory below and calls
 * it; when no such file exists -- or the file is not a plugin of THIS library -- it falls back to the plugins compiled into the library
 * (whitted, directlighting, path; emission, single; stratified, lowdiscrepancy, random; kdtree, grid).
 * The factories are named PbrtHipCreate<Kind>, NOT Create<Kind>: a SearchPath that points at a pbrt-v1 install holds stratified.so,
 * kdtree.so, ... whose Create<Kind>(const ParamSet &, ...) is another ABI; such an object fails to load here (its core symbols are
 * unresolved) or lacks PbrtHipCreate<Kind>, either way a warning at most and the built-in answers.  PBRT_SEARCHPATH (the reference's
 * install) is deliberately not searched.  A plugin may export `int PbrtHipPluginAbi(void)` returning PBRT_HIP_PLUGIN_ABI; a mismatch is refused.
 * What differs from the reference: a factory returns a FLAT DESCRIPTOR (which device kernel family runs, with which parameters) instead of a
 * C++ object with virtual methods -- the work itself happens in HIP kernels behind include/pbrt_hip.h, a plugin cannot add device code.
 * A plugin reads its parameters through the accessor table it is handed (no link-time dependency on libpbrt_host.so); every parameter it
 * looks up counts as used for the "unused parameter" warning of ParamSet::ReportUnused (paramset.cpp:330-346).
 * Return 0 on success; anything else makes the directive fail with the reference's "Unable to load plugin" Error. */
#ifndef PBRT_HIP_PLUGIN_H
#define PBRT_HIP_PLUGIN_H
#include "pbrt_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

#define PBRT_HIP_PLUGIN_ABI 2                              /* 1: round 4 (factories named Create<Kind>), 2: PbrtHipCreate<Kind> */

typedef struct PbrtHipParams PbrtHipParams;                 /* the directive's ParamSet (opaque) */
typedef struct PbrtHipParamsApi {                           /* ParamSet::FindOne* (core/paramset.h:88-120) */
    int (*find_int)(const PbrtHipParams *, const char *name, int dflt);
    float (*find_float)(const PbrtHipParams *, const char *name, float dflt);
    int (*find_bool)(const PbrtHipParams *, const char *name, int dflt);
    const char *(*find_string)(const PbrtHipParams *, const char *name, const char *dflt);    /* every string handed out stays valid until the factory returns */
} PbrtHipParamsApi;

typedef struct PbrtHipSurfaceIntegrator { int32_t kind /* RT_INTEGRATOR_* */, max_depth, strategy /* RT_STRATEGY_* (directlighting) */; } PbrtHipSurfaceIntegrator;
typedef struct PbrtHipVolumeIntegrator { int32_t kind /* RT_VOLUME_* */; float step_size; } PbrtHipVolumeIntegrator;
typedef struct PbrtHipSampler { int32_t kind /* RT_SAMPLER_* */, xsamples, ysamples, jitter, pixelsamples; uint32_t seed; } PbrtHipSampler;
typedef struct PbrtHipAccelerator { RtAccelParams params; } PbrtHipAccelerator;

/* the symbols a plugin exports (one of them per shared object, as in the reference) */
typedef int (*PbrtHipCreateSurfaceIntegratorFn)(const PbrtHipParams *, const PbrtHipParamsApi *, PbrtHipSurfaceIntegrator *out);   /* "PbrtHipCreateSurfaceIntegrator" */
typedef int (*PbrtHipCreateVolumeIntegratorFn)(const PbrtHipParams *, const PbrtHipParamsApi *, PbrtHipVolumeIntegrator *out);     /* "PbrtHipCreateVolumeIntegrator" */
typedef int (*PbrtHipCreateSamplerFn)(const PbrtHipParams *, const PbrtHipParamsApi *, PbrtHipSampler *out);                       /* "PbrtHipCreateSampler" */
typedef int (*PbrtHipCreateAcceleratorFn)(const PbrtHipParams *, const PbrtHipParamsApi *, PbrtHipAccelerator *out);               /* "PbrtHipCreateAccelerator" */

#ifdef __cplusplus
}
#endif
#endif /* PBRT_HIP_PLUGIN_H */
