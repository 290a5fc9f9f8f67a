// rt_device.h -- device-visible scene / frame descriptors (HBM layout; see DESIGN.md "Data layout")
#pragma once
#include "rt_math.h"
#include "../../../include/pbrt_hip.h"

namespace rt {

// One triangle = three 16-byte vectors (48 B, one 128-B line holds 2.67 of them):
//   q0 = p1.x p1.y p1.z e1.x | q1 = e1.y e1.z e2.x e2.y | q2 = e2.z, bits(material | flags<<16), bits(light), 0
//   with e1 = p2 - p1, e2 = p3 - p1 (the first two operations of Triangle::Intersect, done once on the host)
struct DevTri { float4 q0, q1, q2; };
// a quadric occupies one primitive slot: bits has RT_PRIM_QUADRIC set and q0.x holds the index into DevScene::quadrics
#define RT_PRIM_QUADRIC (1u << 17)
struct DevQuadric {          // Sphere (shapes/sphere.cpp:89-99): transforms as Transform::m / ::mInv, row-major
    float w2o[16], o2w[16];
    float radius, zmin, zmax, theta_min, theta_max, phi_max;
    int type, pad;
    float p1[3], p2[3], a, c;   // hyperboloid (hyperboloid.cpp:46-70)
};

// per-vertex shading data of one triangle (RtTriShading) plus what Triangle::Intersect derives from its uvs (trianglemesh.cpp:248-268)
#define RT_PRIM_SHADING (1u << 18)      // tri_shade bits: the triangle has per-vertex N and / or S (DevScene::tri_shading)
struct DevTriShading {
    unsigned flags, xform;
    float uv[6];
    float dpdu[3];             // the geometric dpdu of Triangle::Intersect with this triangle's uvs (not normalised)
    float n[9], s[9];
    float pad[3];
};

struct DevMaterial {
    int type;
    float r[3];      // Kd / Kr
    float t[3];      // Kt
    float on_a, on_b;  // Oren-Nayar A,B (reflection.h:268-277); on_b < 0 => Lambertian
    float ior;
    int has_r, has_t;  // glass.cpp:56-61: a lobe exists only if its colour is not black
    float ks[3];       // plastic: Microfacet reflectance
    float exponent;    // plastic / uber: Blinn exponent = 1/roughness, capped at 1000 (reflection.h:313)
    float kr[3];       // uber: SpecularReflection reflectance (Fresnel 1.5 / 1)
    int has_g, has_kr; // uber: glossy / specular-reflection lobes present (non-black, uber.cpp:71-86)
};

struct DevLight {
    int type;
    float color[3];
    float pos[3];
    int n_samples;
    unsigned first_tri, n_tris;
    int reverse_orientation, flip_normal;
    float area;        // ShapeSet::area / Triangle::Area() (shape.h:123-131, trianglemesh.cpp:329-335)
    float dir[3];      // distant
    int quadric;       // area light on a quadric: index into DevScene::quadrics, else -1 (read by the EXT kernels)
    float w2l[9];      // spot: WorldToLight 3x3
    float cos_total, cos_falloff;
};

#define RT_MAX_DIM_REQ 12         // requests the frame descriptor itself carries (PathIntegrator: 11 one-dimensional, 9 two-dimensional); DirectLighting "all"
                                  // has three per light, without bound (directlighting.cpp:39-66): DevFrame::light_dims
struct DimReq { unsigned f_base, u_base; unsigned short n; unsigned short dims; };

struct DevScene {
    const DevTri *tris;
    const DevQuadric *quadrics;  // read only by the EXT kernels
    const float4 *tri_shade;   // per triangle 2 x float4, read once per HIT: {nn.xyz, material|flip<<16} {sn.xyz, area-light index}:
                               // the geometric normal and BSDF tangent are constants of the triangle, precomputed on the host with
                               // the reference's expressions (rt_shade.h tri_frame) instead of two normalisations per vertex
    const uint2 *nodes;
    const uint2 *tnodes;       // the flat traversal round in index form: the same nodes with the leaves in ENTRY form (rt_traverse.h RT_LE_*): word 0 =
                               // position of the first primitive's record in `ltris` << 2 | 3, word 1 = that entry's flags | the cursor
    const uint4 *tpairs;       // the same tree as sibling PAIRS for the flat traversal round: {below.x, below.y, above.x, above.y}; an interior
                               // node's word 1 is the index of its children's pair, a leaf is in entry form as in `tnodes`.  Both children
                               // arrive with ONE 16-byte gather, and because the traversal carries node CONTENTS (in registers and in the
                               // stack entries) instead of indices, a leaf and a popped subtree root need no fetch of their own: gathers per
                               // ray = interior nodes visited, not nodes visited (-44 % at 1 M triangles).  Pairs are stored depth-first.
    unsigned root_x, root_y;   // contents of the root in that encoding
    unsigned top_pairs, leaf_runs;  // leaf_runs: the leaves own runs of consecutive records (tiny scenes; rt_traverse.h RT_LE_*) instead of entries.
                               // The first top_pairs records of tpairs are the owner blocks of the tree's top levels, breadth-first and packed without
                               // padding (pair_blocks_order; an LDS copy of them was measured and not kept: profiles/r05_lds_top_experiment.patch)
    const float4 *ltris;       // ONE 48-byte record per primitive (p1, e1, e2 as in DevTri, the primitive's index in q2.w), RT_TRI_STRIDE float4 units apart,
                               // in the order the depth-first leaf walk first meets the primitives (round 6; rounds 2-5: one copy per leaf reference)
    const unsigned *lrefs;     // entries (position | RT_LE_* flags) of the third and later primitives of the leaves that hold three or more
    const unsigned *leaf_refs;
    const int *tri_shading_idx;      // [n_tris] index into tri_shading or -1 (EXT kernels; null when no mesh has N / S)
    const DevTriShading *tri_shading;
    const float *xforms;             // [n][32] ObjectToWorld m, mInv of the meshes with N / S
    const DevMaterial *materials;
    const DevLight *lights;
    const float *light_tris;   // [n][16]: 9 vertex floats, per-triangle area, area CDF, pad, emitter normal nl.xyz (flipped), pad
    unsigned n_tris, n_lights;
    float bounds[6];
    // uniform grid (RT_ACCEL_GRID): `nodes` holds one {offset,count} voxel per cell, `leaf_refs` the primitive lists
    int accel_kind;
    int nvox[3];
    float gwidth[3], ginv_width[3];
    RtCamera cam;
    RtVolume vol;
};

#define RT_INTEG_DIRECT_WEIGHTED 3    // device-side template value only: DirectLighting with strategy "weighted" (render_kernel family g_render_kernels_weighted)

struct DevFrame {
    int integrator, max_depth, strategy, volume_integrator;
    float step_size;
    int sampler, xs, ys, jitter, spp;
    unsigned seed;
    int x_pixel_start, y_pixel_start, x_pixel_count, y_pixel_count;
    int x_start, x_end, y_start, y_end;
    float fxw, fyw, inv_fxw, inv_fyw;
    const float *filter_table;   // 256 floats in HBM (L1/L2 resident)
    float *accum;                // 5 planes
    float4 *samples;             // per-shard sample buffer, 2 x float4 per work item, laid out by sample_slot()
    int shard_index, shard_count, tile_pixels;
    int tile_w, tile_h, tiles_x;   // 2-D tiles (RtRenderDesc.tile_pixels < 0): tile_w x tile_h pixel blocks of the sample extent, tiles_x of them per row;
                                   // tile_pixels = tile_w * tile_h then.  tile_w == 0: tiles of tile_pixels consecutive pixels in scanline order
    int dbg_x, dbg_y;            // -DRT_DEBUG_PIXEL builds: print the vertices of the samples of this pixel
    int exit_thresh;             // leave the shared traversal loop when <= this many lanes still traverse (0 = never)
    int phase_sync;             // path integrator: alternate the two halves of the state machine between sweeps (rt_integrate.h)
    int high_occupancy;          // host-side choice of the 5-waves/SIMD kernel flavour (not read by the device)
    int leaf_min;                // batched rounds: keep testing leaf primitives while at least this many lanes hold an untested one
    int mega_tile;               // megakernel, one shard: the work counter hands the samples out in square tiles of this many pixels (0: scanline order); rt_integrate.h tile_order_to_sample
    int xcd_bands;               // megakernel: 8 bands of the work list with their own counters (work_counter[8 * band]), a wave draws from its XCD's band first
    int trav_mode;               // 2 = the flat traversal round (trace_round), 3 = lock-step rounds with pooled leaf tests (tiny trees with fat leaves; natural-allocation kernels only)
    int pipeline;                // host-side choice: the queue pipeline (rt_pipeline.h) instead of the megakernel (not read by the device)
    unsigned long long total_work;     // samples this shard renders
    unsigned long long total_pixels;   // pixels in the sample extent
    // sampler dimension table (Sample::oneD/twoD, sampling.cpp:41-70)
    int n1d, n2d;
    unsigned lhs_total;          // draws consumed by LatinHypercube per sample
    unsigned pixgen_draws;       // draws consumed when a new pixel's strata are generated
    DimReq one_d[RT_MAX_DIM_REQ];
    DimReq two_d[RT_MAX_DIM_REQ];
    const DimReq *light_dims;    // DirectLighting "all" (UniformSampleAllLights): per light {light sample 2-D, BSDF sample 2-D, BSDF component 1-D} in HBM, any number of lights
    int dims_max_n;              // (host) the largest sample count of a 2-D request
    // scratch
    unsigned long long *work_counter;
    unsigned long long *counters;   // RtCounters as 8 u64
    uint2 *spill;                   // traversal-stack overflow  [entry][thread]
    float *frames;                  // specular recursion frames [frame][field][thread]
    float *vol_rays, *vol_state, *vol_samp;   // volume scratch (rt_integrate.h), null without a volume
    int vol_nmax;
    unsigned n_threads;
    // DirectLighting "weighted" (WeightedSampleOneLight transport.cpp:71-122; rt_weighted.h): the frame runs as three passes of the megakernel
    int weighted_phase;             // 0: not a weighted frame, 1: count the shading points of every sample, 2: survey (every light's estimate), 3: the frame proper
    unsigned *wt_base;              // [total_work + 1]: pass 1 leaves a sample's shading-point count at [work]; scanned in place to the first ordinal
    float *wt_rec;                  // [point][1 + 2 * n_lights]: the light-number sample, then per light y(Ld) and y(n_lights * Ld)       (pass 2)
    float2 *wt_pick;                // [point]: the chosen light (int bits) and lightSampleWeight, 0 = the uniform start-up branch       (recurrence kernel)
    // lights of MIXED RNG use (an emitter of several triangles draws its triangle, ShapeSet::Sample shape.h:115-121; every other light draws nothing): the RNG
    // counter at a sample's j-th shading point depends on how many of the j points before it chose a drawing light (k = 0 .. j), so the survey keeps, for
    // every drawing light, one estimate per k; a point's record is [u | per light: (y, y*n) once, or j + 1 times for a drawing light] and the records of
    // sample `work` start at wt_recbase[work]
    int wt_mixed;
    unsigned wt_nd;                 // number of drawing lights
    const unsigned *wt_recbase;     // [total_work + 1], float offsets into wt_rec
#ifdef RT_TAIL_PROBE
    unsigned long long *probe;      // -DRT_TAIL_PROBE builds (tools/build_variant.py): per megakernel wave {start, work list found empty, end} in 10 ns ticks + samples taken
#endif
};

// Explicit address spaces.  Pointers that live inside DevScene/DevFrame are loaded from memory, so the compiler cannot
// infer where they point and emits FLAT loads (which tie up both the vector-memory and the LDS counters); every hot
// dereference therefore goes through one of these: global_load_* for HBM data, ds_* for the traversal stack.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_G __attribute__((address_space(1)))
#define RT_L __attribute__((address_space(3)))
#else
#define RT_G
#define RT_L
#endif
typedef const DevMaterial RT_G &MatRef;
typedef const DevLight RT_G &LightRef;
#define RT_MAT(sc, i) (*((const DevMaterial RT_G *)(sc).materials + (i)))
#define RT_LIGHT(sc, i) (*((const DevLight RT_G *)(sc).lights + (i)))
#define RT_GPTR(T, p) ((T RT_G *)(p))

// Sample-buffer layout (round 3).  Work item w of a shard is sample s = w % spp of the shard's local pixel lp = w / spp (local pixels
// follow the shard's tiles in order, so on one rank lp is the pixel's scanline index).  64 consecutive local pixels form a chunk stored
// [sample][pixel in chunk][L.rgb+alpha | imageX, imageY, -, -]: the lanes of a film-gather wave, which sit on consecutive pixels of a
// row and read the same sample slot of each, fetch 2 KB of consecutive records per load pair instead of 64 separate lines, and the
// render kernel writes a sample's two float4s into one 32-byte sector.
// Returns the float4 index of the L record; the image-position record is RT_SAMPLE_XY float4s further.
#define RT_SAMPLE_XY 1
__host__ __device__ inline unsigned long long sample_slot(unsigned lp, unsigned s, int spp) {
    return ((unsigned long long)(lp >> 6) * unsigned(spp) + s) * 128ull + (lp & 63u) * 2u;
}

}  // namespace rt
