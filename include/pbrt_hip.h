/* pbrt_hip.h -- C ABI of the MI355X (gfx950) implementation of pbrt-v1's
 * Scene::Render hot path.  Plain C: pointers, sizes and PODs only.
 *
 * The reference has NO foreign-function interface for this path: Scene::Render
 * (core/scene.cpp:32-88) is a non-virtual C++ loop that calls C++ plugin objects
 * created by extern "C" factories taking C++ types (core/dynload.cpp:112-260).  The
 * boundary below is therefore the one a maintainer would introduce between
 * RenderOptions::MakeScene (core/api.cpp:484-529) and Scene::Render
 * (core/api.cpp:475): everything MakeScene has produced is handed over as flat
 * arrays, the frame is rendered on the GPU, and the film comes back in the layout
 * ImageFilm keeps (film/image.cpp:57-65).  INTEGRATION.md shows the reference-side
 * stub.  Each entry point cites the reference interface it replaces.
 *
 * Conventions: every function returns 0 on success, a negative RT_E* code on
 * failure (no exceptions cross the boundary, like the reference);
 * rt_last_error() gives the message.  The caller owns all host buffers; the
 * library owns device memory.  One hipStream_t per handle; a handle may be used
 * from one thread at a time.  All arithmetic is IEEE float32 compiled without
 * FMA contraction so that results track the reference's SSE build.
 */
#ifndef PBRT_HIP_H
#define PBRT_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_EINVAL (-1)   /* bad argument / inconsistent description          */
#define RT_EDEVICE (-2)  /* HIP runtime error (no device, OOM, launch failed) */
#define RT_ESTATE (-3)   /* call out of order (e.g. render before film bind)  */
#define RT_ENOMEM (-4)   /* host memory exhausted while building the scene    */

typedef struct RtScene RtScene; /* opaque */

/* ---- materials: materials/matte.cpp:46-64, glass.cpp:46-63, mirror.cpp:42-55, plastic.cpp:47-69, uber.cpp:52-89 ---- */
enum { RT_MAT_MATTE = 0, RT_MAT_MIRROR = 1, RT_MAT_GLASS = 2, RT_MAT_PLASTIC = 3, RT_MAT_UBER = 4 };
typedef struct RtMaterial {
    int32_t type;
    float kd[3];   /* matte/plastic Kd  | mirror/glass Kr  (already .Clamp()'ed >= 0) */
    float kt[3];   /* glass Kt                                                   */
    float sigma;   /* matte: Oren-Nayar sigma in degrees, clamped [0,90]; 0 = Lambertian */
    float ior;     /* glass "index"                                              */
    float ks[3];   /* plastic Ks (.Clamp()'ed): Microfacet(Ks, FresnelDielectric(1.5, 1), Blinn(1/roughness)) */
    float roughness; /* plastic / uber "roughness"                                */
    float kr[3];   /* uber: op*Kr (SpecularReflection, FresnelDielectric(1.5, 1)).  For uber kd = op*Kd, ks = op*Ks,
                    * kt = 1 - op (SpecularTransmission(1 - op, 1, 1), present iff op != 1), ior = 1; lobe order T, D, G, R */
} RtMaterial;

/* ---- lights: lights/point.cpp:49-69, lights/area.cpp:28-105, lights/spot.cpp:54-79, lights/distant.cpp:51-62 ---- */
enum { RT_LIGHT_POINT = 0, RT_LIGHT_AREA = 1, RT_LIGHT_SPOT = 2, RT_LIGHT_DISTANT = 3 };
typedef struct RtLight {
    int32_t type;
    float color[3];      /* point/spot: I   area: Lemit   distant: L                 */
    float pos[3];        /* point/spot light position (world, LightToWorld(0,0,0))  */
    int32_t n_samples;   /* Light::nSamples (light.h:39)                            */
    uint32_t first_tri;  /* area: range in light_tris (ShapeSet order, shape.h:112) */
    uint32_t n_tris;
    int32_t reverse_orientation; /* Triangle::Sample flips Ns by this only (trianglemesh.cpp:346) */
    int32_t flip_normal;         /* reverseOrientation ^ transformSwapsHandedness (shape.cpp:49)   */
    float dir[3];                /* distant: lightDir = Normalize(LightToWorld(from - to)) (distant.cpp:51-56) */
    float world_to_light[9];     /* spot: upper-left 3x3 of WorldToLight, row-major (Falloff, spot.cpp:68-79)   */
    float cos_total_width, cos_falloff_start;   /* spot.cpp:58-59 */
    int32_t quadric_plus1;       /* area light whose shape is a quadric (AreaLight keeps a CanIntersect shape as is,
                                  * area.cpp:38-39): 1 + index into RtSceneDesc::quadrics, n_tris = 0; 0 = triangle set */
} RtLight;

/* ---- camera: core/camera.cpp:50-70, cameras/perspective.cpp:51-82, orthographic.cpp:48-79, environment.cpp:47-61 ---- */
enum { RT_CAMERA_PERSPECTIVE = 0, RT_CAMERA_ORTHOGRAPHIC = 1, RT_CAMERA_ENVIRONMENT = 2 };
typedef struct RtCamera {
    float raster_to_camera[16]; /* row-major 4x4, as Matrix4x4::m (core/transform.h) */
    float camera_to_world[16];
    float lens_radius, focal_distance, hither, yon;
    float shutter_open, shutter_close;
    int32_t type;                /* RT_CAMERA_*                                                        */
    int32_t x_res, y_res;        /* film resolution (environment camera: theta/phi from imageY/imageX) */
} RtCamera;

/* ---- homogeneous medium: volumes/homogeneous.cpp:27-74 ---- */
typedef struct RtVolume {
    int32_t present;
    float world_to_volume[16];
    float p0[3], p1[3];
    float sigma_a[3], sigma_s[3], le[3];
    float g;
} RtVolume;

/* ---- accelerator parameters: accelerators/kdtree.cpp:489-498, grid.cpp:431-434 ---- */
enum { RT_ACCEL_KDTREE = 0, RT_ACCEL_GRID = 1 };
typedef struct RtAccelParams {
    int32_t kind;
    int32_t isect_cost, trav_cost, max_prims, max_depth; /* kd defaults 80, 1, 1, -1 */
    float empty_bonus;                                    /* kd default 0.5           */
    int32_t build_threads;                                /* 0 = auto; -1 = the checking form of the kd build: one thread, every node
                                                           * sorts its own edges as kdtree.cpp:246 does (same arrays, by test)  */
} RtAccelParams;

/* ---- quadrics: shapes/sphere.cpp:89-215, disk.cpp:51-130, cylinder.cpp:52-180, cone.cpp:41-186, paraboloid.cpp:44-190,
 * hyperboloid.cpp:46-240.  A quadric is ONE primitive of the accelerator
 * (Shape::CanIntersect, primitive.cpp:40-53).  In the primitive arrays below it occupies one slot whose tri_verts
 * hold its world bound as a degenerate triangle {pMin, pMax, pMin} (Shape::WorldBound shape.h:57-59) and whose
 * tri_flags has bit1 set; the k-th such slot is quadrics[k]. ---- */
enum { RT_QUADRIC_SPHERE = 0, RT_QUADRIC_DISK = 1, RT_QUADRIC_CYLINDER = 2, RT_QUADRIC_CONE = 3, RT_QUADRIC_PARABOLOID = 4, RT_QUADRIC_HYPERBOLOID = 5 };
typedef struct RtQuadric {
    int32_t type;
    float object_to_world[16], world_to_object[16]; /* Transform::m / ::mInv, row-major                      */
    float radius, zmin, zmax, theta_min, theta_max, phi_max; /* as the ctors store them (sphere.cpp:89-99, cylinder.cpp:52-59);
                                                              * disk (disk.cpp:51-59): zmin = height, zmax = innerRadius;
                                                              * cone (cone.cpp:41-48): zmin = 0, zmax = height; paraboloid (paraboloid.cpp:44-52);
                                                              * hyperboloid (hyperboloid.cpp:46-70): radius = rmax + the fields below */
    float p1[3], p2[3], a, c;                                /* hyperboloid: end points (after the ctor's swap) and the implicit form's a, c */
} RtQuadric;

/* ---- per-vertex shading data of triangle meshes: "uv" / "st", "N", "S" (shapes/trianglemesh.cpp:71-133 GetShadingGeometry,
 * :248-274 the (u,v)-dependent dpdu / dpdv of Triangle::Intersect, :315-328 GetUVs, checks of the factory :350-406).  A triangle
 * whose mesh has any of them refers to one record.  n and s are object-space (the mesh keeps them so, :155-164) and go through
 * the mesh's ObjectToWorld at shading time: xform indexes RtSceneDesc::xforms. ---- */
enum { RT_SHADING_UV = 1, RT_SHADING_N = 2, RT_SHADING_S = 4 };
typedef struct RtTriShading {
    uint32_t flags;            /* RT_SHADING_*                                                               */
    uint32_t xform;
    float uv[6];               /* uvs[3][2] as GetUVs returns them (the defaults (0,0) (1,0) (1,1) without "uv") */
    float n[9];                /* normals of the three vertices, object space                                */
    float s[9];                /* tangents of the three vertices, object space                               */
} RtTriShading;

/* Scene description = what MakeScene hands to Scene::Scene (core/scene.cpp:100-119),
 * flattened.  Triangles are in the order KdTreeAccel's FullyRefine produces
 * (kdtree.cpp:146-148: per mesh, last triangle first). Vertices are world space
 * (trianglemesh.cpp:166-168). */
typedef struct RtSceneDesc {
    uint32_t n_tris;
    const float *tri_verts;       /* [n_tris][9]  p1 p2 p3                                   */
    const uint16_t *tri_material; /* [n_tris] index into materials                             */
    const int32_t *tri_light;     /* [n_tris] area-light index (GetAreaLight) or -1           */
    const uint8_t *tri_flags;     /* [n_tris] bit0 = flip geometric normal (shape.cpp:49-50), bit1 = quadric slot */
    uint32_t n_materials;
    const RtMaterial *materials;
    uint32_t n_lights;
    const RtLight *lights;
    uint32_t n_light_tris;
    const float *light_tris;      /* [n_light_tris][9] emitter triangles, ShapeSet order       */
    RtCamera camera;
    RtVolume volume;
    RtAccelParams accel;
    uint32_t n_quadrics;          /* may be 0 / NULL                                           */
    const RtQuadric *quadrics;
    const int32_t *tri_shading;   /* [n_tris] index into shading[] or -1; NULL when no mesh has uv / N / S */
    uint32_t n_shading;
    const RtTriShading *shading;
    uint32_t n_xforms;
    const float *xforms;          /* [n_xforms][32]: Transform::m then ::mInv of a mesh's ObjectToWorld, row-major */
} RtSceneDesc;

/* ---- per-frame description ---- */
enum { RT_INTEGRATOR_WHITTED = 0, RT_INTEGRATOR_DIRECT = 1, RT_INTEGRATOR_PATH = 2 };
enum { RT_STRATEGY_ALL = 0, RT_STRATEGY_ONE = 1, RT_STRATEGY_WEIGHTED = 2 };
/* RT_STRATEGY_WEIGHTED: WeightedSampleOneLight (transport.cpp:71-122), a recurrence over every shading point of the frame in program order.
 * rt_render accepts it on one shard (shard_count == 1) with at most 2048 lights of any mix (round 5: emitters of several triangles, which draw one
 * random number per estimate -- ShapeSet::Sample, shape.h:115-121 -- next to lights that draw none: the survey then keeps one estimate per possible
 * position of that draw in the sample's stream).
 * Such a frame is five launches (RtRenderStats.weighted_ms); rt_render waits on the stream once in the middle of it, for the number of shading
 * points that sizes the survey's tables (every other frame is asynchronous from the first launch on).  That wait is also the one place where
 * rt_render can fail AFTER launching work (2^32 or more shading points, no memory for the survey's tables): the film and the sample buffer are
 * untouched then (the count pass writes neither), rt_last_render_stats / rt_samples_read report "no frame".  In a participating medium the
 * strategy needs lights that draw no random numbers at all (delta lights, single-triangle / quadric emitters): the medium makes an estimate's
 * draws depend on occlusion, and an emitter of several triangles would draw its triangle from a different place in the stream than the
 * survey did -- refused.
 * Cost and limit with lights of mixed RNG use: a camera sample with P shading points keeps P (1 + 2 (nLights - nD)) + nD P (P + 1) floats of survey records
 * (nD = emitters of several triangles) -- QUADRATIC in P, i.e. in the depth of the specular recursion -- capped at 2^32 floats per frame (refused after the
 * count pass, as above), and the recurrence itself is sequential by definition: one lane walks every shading point of the frame in program order (~0.45 us per
 * point on MI355X; the reference pays the same dependence on one core).  High maxdepth with many such emitters is therefore slow or refused, never wrong. */
enum { RT_VOLUME_NONE = 0, RT_VOLUME_EMISSION = 1, RT_VOLUME_SINGLE = 2 };
enum { RT_SAMPLER_STRATIFIED = 0, RT_SAMPLER_LOWDISCREPANCY = 1, RT_SAMPLER_RANDOM = 2 };

typedef struct RtRenderDesc {
    /* SurfaceIntegrator: integrators/{whitted,directlighting,path}.cpp factories */
    int32_t integrator, max_depth, strategy;
    /* VolumeIntegrator: integrators/{emission,single}.cpp */
    int32_t volume_integrator;
    float step_size;
    /* Sampler: samplers/{stratified,lowdiscrepancy,random}.cpp */
    int32_t sampler, x_samples, y_samples, jitter, pixel_samples;
    uint32_t seed;            /* counter-RNG seed (see oracle/ref/keyed_rng.cpp)         */
    /* Film: film/image.cpp:69-101,148-156 */
    int32_t x_res, y_res;
    int32_t x_pixel_start, y_pixel_start, x_pixel_count, y_pixel_count; /* crop window in pixels */
    int32_t x_start, x_end, y_start, y_end;  /* sample extent (GetSampleExtent)               */
    float filter_x_width, filter_y_width;
    float filter_table[256];  /* 16x16, image.cpp:89-101                                   */
    /* Work partition (reference: cropwindow processes, image.cpp:220-228): pixels of the
     * sample extent, in scanline order, are grouped into tiles of tile_pixels consecutive
     * pixels; this call renders tiles t with t % shard_count == shard_index.
     * tile_pixels < 0 selects 2-D tiles: -tile_pixels = tile_w | tile_h << 16, blocks of tile_w x tile_h pixels of the sample extent
     * numbered row by row (border blocks are clipped).  A rank then owns compact pieces of the film: its film gather touches only the
     * blocks around them, and the partial films can be merged row-wise (bench.py: reduce-scatter + per-rank resolve).
     * Border blocks are whole blocks whose pixels outside the extent are fetched and dropped (a lane idles for about a ray's time per dropped
     * item): pick sizes that pad the extent little (pbrt-v1_amd ParsedScene.set_shard(fit=True)).  With shard_count == 1 the tiles partition
     * nothing and a frame without a medium is rendered in scanline order whatever is passed here. */
    int32_t shard_index, shard_count, tile_pixels;
} RtRenderDesc;

/* per-frame counters (reference: StatsCounters, core/util.cpp:186-285) */
typedef struct RtCounters {
    uint64_t camera_rays;     /* "Camera Rays Traced" scene.cpp:80            */
    uint64_t closest_rays;    /* every Scene::Intersect  (camera+bounce+MIS)  */
    uint64_t any_rays;        /* every Scene::IntersectP ("shadow rays", light.cpp:32) */
    uint64_t nodes_visited;   /* kd nodes / grid voxels touched               */
    uint64_t leaf_refs;       /* leaf primitive references read               */
    uint64_t tri_tests;       /* Triangle::Intersect(P) calls, trianglemesh.cpp:217 */
    uint64_t bad_samples;     /* NaN/negative/inf radiance, scene.cpp:60-74   */
    uint64_t stack_overflows; /* traversal stack entries spilled beyond LDS   */
} RtCounters;

typedef struct RtRay { float o[3], d[3], mint, maxt; } RtRay;             /* geometry.h:204-217 */
typedef struct RtHit { int32_t prim; float t, b1, b2; } RtHit;            /* prim < 0 : miss      */
typedef struct RtAccelInfo {
    uint32_t n_nodes, n_leaf_refs, max_depth, n_tris; /* grid: n_nodes = voxels, n_leaf_refs = voxel list entries */
    float bounds[6];
    double build_seconds;
    int32_t kind;                                       /* RT_ACCEL_KDTREE / RT_ACCEL_GRID */
    int32_t grid_nvoxels[3];                            /* GridAccel::NVoxels (grid.cpp:146-152) */
    float grid_width[3], grid_inv_width[3];             /* GridAccel::Width / InvWidth (grid.cpp:154-158) */
} RtAccelInfo;

const char *rt_last_error(void);
int rt_device_count(int *count);

/* Build accelerator (KdTreeAccel ctor kdtree.cpp:141-190 / GridAccel ctor grid.cpp:122-210),
 * upload everything.  device < 0 = current device. */
int rt_scene_create(const RtSceneDesc *desc, int device, RtScene **out);
/* The same with an accelerator built elsewhere -- by rt_accel_build, or copied out of another process' scene (rt_scene_accel_info /
 * rt_scene_accel_copy): the ranks of one node build the tree once (reference: every cropwindow process of image.cpp:220-228 builds its
 * own, kdtree.cpp:141-312).  Arrays as rt_accel_copy writes them; they are checked (every index in range) and copied. */
typedef struct RtPrebuiltAccel {
    int32_t kind;                       /* RT_ACCEL_KDTREE / RT_ACCEL_GRID, must equal desc->accel.kind */
    uint32_t n_nodes, n_leaf_refs, max_depth;
    const uint32_t *nodes;              /* [n_nodes][2] */
    const uint32_t *leaf_refs;          /* [n_leaf_refs] */
    float bounds[6];
    int32_t grid_nvoxels[3]; float grid_width[3], grid_inv_width[3];   /* grid only (RtAccelInfo) */
} RtPrebuiltAccel;
int rt_scene_create_prebuilt(const RtSceneDesc *desc, int device, const RtPrebuiltAccel *accel, RtScene **out);
int rt_scene_destroy(RtScene *s);
int rt_scene_set_stream(RtScene *s, void *hip_stream);
int rt_scene_accel_info(const RtScene *s, RtAccelInfo *info);
/* copy the flattened kd-tree back (tests / oracle traverse the same tree):
 * nodes = [n_nodes][2] u32, leaf_refs = [n_leaf_refs] u32 */
int rt_scene_accel_copy(const RtScene *s, uint32_t *nodes, uint32_t *leaf_refs);

/* The accelerator build alone, host-only (no device needed): KdTreeAccel's constructor,
 * accelerators/kdtree.cpp:141-312.  rt_scene_create runs exactly this. */
/* Either accelerator, host-only: kd-tree as above, or GridAccel's constructor in its "refineimmediately" form
 * (accelerators/grid.cpp:122-210).  nodes = [n_nodes][2] u32: kd nodes, or per-voxel {offset, count}. */
typedef struct RtKdTree RtAccel;
int rt_accel_build(const float *tri_verts, uint32_t n_tris, const RtAccelParams *params, RtAccel **out);
int rt_accel_info(const RtAccel *t, RtAccelInfo *info);
int rt_accel_copy(const RtAccel *t, uint32_t *nodes, uint32_t *leaf_refs);
/* Host only, for tests of the host logic: the leaves of a kd-tree as rt_scene_create lays them out for the flat traversal -- ONE 48-byte record per primitive, `stride` float4
 * units apart, in the order the depth-first leaf walk first meets the primitives (`slot_prim[slot]` = primitive), and per leaf node its primitives as ENTRIES = position | flags
 * (the reference keeps index lists, accelerators/kdtree.cpp:55-64): word 0 = position of the first primitive's record << 2 | 3, word 1 = bit 31 "more follow",
 * bit 30 "the others' entries are a list" | the second entry (leaf of two) or HALF the index of the list in `entries`, whose items carry the same two flags; an empty leaf's first entry
 * is 0xffffffff.  `runs` != 0: every leaf owns a run of consecutive records, word 1 = bit 31 (more than one) | the count (what scenes of a few thousand references get); `copies` != 0:
 * one record per leaf reference (measurements).  Call with null arrays for the sizes (tnodes [n_nodes][2], slot_prim [n_slots], entries [n_entries]). */
typedef struct RtLeafLayoutInfo { uint64_t n_nodes, n_slots, n_entries; uint32_t stride, pad; } RtLeafLayoutInfo;
int rt_accel_leaf_layout(const RtAccel *t, int runs, int copies, uint32_t *tnodes, uint32_t *slot_prim, uint32_t *entries, RtLeafLayoutInfo *info);
int rt_accel_destroy(RtAccel *t);
typedef struct RtKdTree RtKdTree;
int rt_kdtree_build(const float *tri_verts, uint32_t n_tris, const RtAccelParams *params, RtKdTree **out);
int rt_kdtree_info(const RtKdTree *t, RtAccelInfo *info);
int rt_kdtree_copy(const RtKdTree *t, uint32_t *nodes, uint32_t *leaf_refs);
int rt_kdtree_destroy(RtKdTree *t);

/* Unit entry points (parity of one stage in isolation) */
/* Camera::GenerateRay for `count` samples starting at camera-sample index `first`
 * (perspective.cpp:51-82 + the sampler's image/lens positions) */
int rt_camera_rays(RtScene *s, const RtRenderDesc *rd, uint64_t first, uint32_t count, RtRay *rays_out);
/* Scene::Intersect (kdtree.cpp:313-403 + trianglemesh.cpp:213-278) */
int rt_trace_closest(RtScene *s, const RtRay *rays, uint32_t n, RtHit *hits_out);
/* Scene::IntersectP (kdtree.cpp:404-488 + trianglemesh.cpp:279-314); occluded_out[i] in {0,1} */
int rt_trace_any(RtScene *s, const RtRay *rays, uint32_t n, uint8_t *occluded_out);

/* Film accumulators, ImageFilm::Pixel (image.cpp:57-65) as planes:
 * float accum[5][y_pixel_count][x_pixel_count] = sum w*L.r, .g, .b, sum w*alpha, sum w.
 * rt_film_bind: accumulate into caller-provided DEVICE memory (e.g. a torch tensor that is
 * later all-reduced by RCCL); pass NULL to let the library allocate. Zeroes nothing. */
int rt_film_bind(RtScene *s, void *device_accum, int32_t x_pixel_count, int32_t y_pixel_count);
int rt_film_clear(RtScene *s);
int rt_film_read(RtScene *s, float *host_accum);          /* 5*W*H floats, synchronises */
/* ImageFilm::WriteImage minus the file: XYZ round trip, /weightSum, clamps, premultiply
 * (image.cpp:157-203).  rgb_out[H][W][3], alpha_out[H][W] host buffers. */
int rt_film_resolve(RtScene *s, int premultiply_alpha, float *rgb_out, float *alpha_out);

/* Scene::Render's sample loop (scene.cpp:42-84) for this shard, asynchronous on the
 * handle's stream.  rt_sync waits.  Counters accumulate until rt_counters_reset. */
int rt_render(RtScene *s, const RtRenderDesc *rd);
int rt_sync(RtScene *s);
/* per-camera-sample results of the last rt_render, before filtering: out[count][8] = L.rgb, alpha, imageX, imageY, 0, 0 in the
 * sampler's order (what Scene::Render hands to Film::AddSample, scene.cpp:76).  The reference-side binding
 * (oracle/ref/hip_adapter.cpp) serves SurfaceIntegrator::Li from it.
 * Order: record w is work item w of this shard's work list.  shard_count == 1 rendered by the megakernel (every frame without a
 * medium): scanline order of the sample extent, whatever tile shape the descriptor names (one shard's tiles partition nothing).
 * Otherwise the shard's tiles in order, a tile's pixels in scanline order, a pixel's samples consecutively; with 2-D tiles the
 * border tiles are padded to whole tiles and the work items that fall off the sample extent are never rendered: their records hold
 * zeros or what an earlier frame left there (finite), and the film gathers never look at them. */
int rt_samples_read(RtScene *s, uint64_t first, uint64_t count, float *out);
int rt_counters(RtScene *s, RtCounters *out);             /* synchronises */
int rt_counters_reset(RtScene *s);
/* per-ray statistics cost a few VALU ops per node visit: enabled (default) for parity/accounting runs,
 * disabled for timed runs; the frame is deterministic so counts of one run describe the other */
int rt_set_counting(RtScene *s, int enabled);
/* elapsed GPU milliseconds of the last rt_render launch sequence (HIP events on the
 * handle's stream) and of its dominant kernel */
int rt_last_render_ms(RtScene *s, float *total_ms, float *kernel_ms);
/* The same by kernel.  Small, cache-resident scenes render with one persistent megakernel (pipeline = 0: trace_ms == render_ms);
 * large ones with the queue pipeline (pipeline = 1): `iterations` alternations of the shade kernel (SurfaceIntegrator::Li between
 * two rays, for every path slot) and the trace kernel (KdTreeAccel::Intersect / IntersectP for the queued rays); trace_ms is the
 * sum over the first `timed_iterations` trace launches (all of them unless a frame needs more than 256). */
typedef struct RtRenderStats {
    float total_ms, render_ms, trace_ms, gather_ms;
    int32_t pipeline, iterations, timed_iterations;
    uint32_t slots;
    float shade_ms;            /* queue pipeline: the shade launches summed */
    int32_t bands;             /* megakernel: 1 (one launch per frame); pipeline: 0 */
    float march_ms;            /* queue pipeline with a medium: the ray-march launches summed (rt::pipe_march_kernel: the marches' shadow rays are traced inside it) */
    /* DirectLighting "weighted" (three megakernel passes + the recurrence; render_ms is all of them): the frame's shading points = calls of
     * WeightedSampleOneLight, and the ms of the count pass, the scan, the survey pass, the recurrence kernel and the frame pass; zeros otherwise */
    uint64_t weighted_points;
    float weighted_ms[5];
} RtRenderStats;
/* ImageFilm::WriteImage's normalisation (image.cpp:157-203) of ANY 5-plane accumulator in device memory (planes of n floats each), e.g. the
 * rows of the film a rank owns after a reduce-scatter; rgb[n][3] and alpha[n] stay on the device.  Asynchronous on the scene's stream. */
int rt_film_resolve_device(RtScene *s, const float *dev_accum, uint64_t n, int premultiply, float *dev_rgb, float *dev_alpha);
/* The same with the result interleaved, dev_rgba[n][4] (16-byte aligned): what ONE all-gather moves when every rank resolves its own rows. */
int rt_film_resolve_device_rgba(RtScene *s, const float *dev_accum, uint64_t n, int premultiply, float *dev_rgba);
/* N > 1 film merge, send side.  The reference has no in-process merge: every cropwindow process writes its own image
 * (film/image.cpp:220-228) and tools/exrassemble.cpp:42-75 adds them up.  Here a rank's full-frame 5-plane film `dev_accum`
 * (h rows of w) is re-laid as `world` parts of `rows` film rows each, part r = [5][rows][w] (rows beyond h zero; world * rows >= h):
 * the send buffer of ONE reduce-scatter whose r-th chunk is everything rank r resolves.  dev_parts = world*5*rows*w floats.
 * Asynchronous on the scene's stream. */
int rt_film_pack_parts(RtScene *s, const float *dev_accum, int32_t w, int32_t h, int32_t world, int32_t rows, float *dev_parts);
int rt_last_render_stats(RtScene *s, RtRenderStats *out);

#ifdef __cplusplus
}
#endif
#endif /* PBRT_HIP_H */
