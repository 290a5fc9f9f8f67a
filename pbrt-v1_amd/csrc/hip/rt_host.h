// rt_host.h -- what the host-side translation units of libpbrt_hip.so share: the scene object behind the opaque RtScene handle, error / knob helpers,
// device-buffer helpers.  rt_scene.hip builds scenes (layouts, uploads, the accelerator ABI), rt_film.hip owns the film kernels and the rt_film_* ABI,
// rt_kernels.hip renders (make_frame, the megakernel / queue-pipeline dispatch, rt_render, rt_trace_*).  Round 6: split out of a 2 300-line rt_kernels.hip.
#pragma once
#include "rt_render_kernel.h"
#include "rt_pipeline.h"
#include "rt_pipe_vertex.h"
#include "rt_pipe_march.h"
#include "rt_internal.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <cmath>
#include <chrono>
#include <atomic>
#include <thread>

namespace rt {
// Experiment / test knobs (PBRT_HIP_*: kernel flavour, pipeline form, film-gather kernel, layout switches, logs) are read only when
// PBRT_HIP_TUNE is set in the environment -- the tests and tools/ set it -- so a production process cannot change its behaviour through a stray
// variable; -DRT_NO_TUNABLES compiles them out.  (rt_kernels.hip)
const char *knob(const char *name);
// status code + message for rt_last_error() (thread-local; rt_kernels.hip)
int fail(int code, const std::string &msg);
const char *last_error();
void hip_warn(hipError_t e, const char *what);
}  // namespace rt
using namespace rt;
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return rt::fail(RT_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)
#define HIPWARN(expr) rt::hip_warn((expr), #expr)

// render_kernel instantiations live in rt_mega_{w,d,p}.hip, 16 per integrator: k = (VOL*2 + ACCEL)*2 + COUNT for the natural-allocation
// kernels (0..7; the counting twins always carry the glossy / quadric code, they are not timed), 8 + VOL*2 + ACCEL for the
// high-occupancy flavour, 12 + VOL*2 + ACCEL for the timed kernels with the glossy (plastic) lobes and quadric slots compiled in
// (EXT: powf and the second lobe cost ~17 VGPRs, one wave per SIMD less for DirectLighting).  `variant` keeps round 1's numbering:
// ((VOL*2 + ACCEL)*2 + COUNT)*3 + INTEG | 24 + (VOL*2 + ACCEL)*3 + INTEG | 36 + (VOL*2 + ACCEL)*3 + INTEG.
namespace rt { extern const RenderKernelFn g_render_kernels_whitted[16], g_render_kernels_direct[16], g_render_kernels_path[16], g_render_kernels_weighted[8]; }
namespace rt { extern const PipeShadeFn g_pipe_shade_whitted[6], g_pipe_shade_direct[6], g_pipe_shade_path[6]; extern const PipeTraceFn g_pipe_trace[8]; extern const PipeShadeFn g_pipe_vertex[3];
               extern const PipeMarchFn g_pipe_march[6]; }
static inline RenderKernelFn render_kernel_of(int variant) {
    const RenderKernelFn *t = (variant % 3 == 0) ? g_render_kernels_whitted : (variant % 3 == 1) ? g_render_kernels_direct : g_render_kernels_path;
    return t[variant < 24 ? variant / 3 : variant < 36 ? 8 + (variant - 24) / 3 : 12 + (variant - 36) / 3];
}

struct RtScene {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    KdTree tree;
    GridAccelData gridacc;
    int accel_kind = RT_ACCEL_KDTREE;
    double per_leaf = -1.0;             // average primitives per non-empty kd leaf (traversal heuristics), computed on first use
    bool has_ext = false;               // plastic materials or quadrics present: use the kernels that carry that code (EXT)
    DevScene dev{};
    std::vector<void *> allocs;
    // film
    float *accum = nullptr; bool own_accum = false; int film_w = 0, film_h = 0;
    float *filter_dev = nullptr;
    // per-launch scratch
    unsigned long long *work_counter = nullptr, *counters = nullptr;
    uint2 *spill = nullptr; size_t spill_entries = 0;
    float *frames = nullptr; size_t frames_floats = 0;
    unsigned grid = 0, n_threads = 0;
    unsigned grids[48] = {0};          // resident grid per render_kernel<COUNT, INTEG> instantiation
    unsigned wgrids[8] = {0};          // ... of the DirectLighting "weighted" family (rt_mega_dw.hip)
    DimReq *light_dims = nullptr; size_t light_dims_cap = 0;      // DirectLighting "all": the per-light sample requests (make_frame)
    std::vector<DimReq> light_dims_host;
    const unsigned *light_draw_flags = nullptr; unsigned n_drawing_lights = 0;
    unsigned *wt_recbase = nullptr; size_t wt_recbase_cap = 0;
    int light_draws = 0;               // RandomFloat()s one EstimateDirect draws: the same for every light (0 / 1), or -1 when the lights differ
    unsigned *wt_base = nullptr; size_t wt_base_cap = 0; float *wt_rec = nullptr; size_t wt_rec_cap = 0; float2 *wt_pick = nullptr; size_t wt_pick_cap = 0;
    unsigned long long *wt_sums = nullptr;                     // per-block sums of the point-count scan
    unsigned long long *wt_total = nullptr;                    // page-locked: the frame's shading points (weighted_scan_top_kernel)
    unsigned long long wt_points = 0; bool last_weighted = false;
    hipEvent_t wt_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    DevScene *dev_scene = nullptr; DevFrame *dev_frame = nullptr;   // descriptors in HBM (read with scalar loads)
    float4 *samples = nullptr; size_t samples_cap = 0;          // per-shard sample buffer
    int samples_spp = 1;
    unsigned long long samples_last = 0;                       // camera samples the LAST rt_render wrote (rt_samples_read's range)
    float ms_render = 0.f, ms_gather = 0.f; hipEvent_t ev2 = nullptr;
    float *resolve_buf = nullptr; size_t resolve_cap = 0;
    float *vol_buf = nullptr; size_t vol_cap = 0;          // volume scratch: rays | state | samp
    RtVolume volume{};
    int spill_depth = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_timing = false;
    bool counting = true;
    uint32_t n_tris = 0;
    size_t n_leaf_tri_units = 0, n_leaf_entries = 0;
    // queue pipeline (rt_pipeline.h)
    PipePool pool{}; unsigned pool_cap = 0; int pool_vec = 0; PipePool *dev_pool = nullptr;
    unsigned *h_qcount = nullptr;                       // page-locked mirror of pool.q_count (termination test)
    unsigned trace_grids[8] = {0}, march_grids[6] = {0};
    std::vector<hipEvent_t> pipe_ev;                    // [6 * RT_PIPE_TIMED]: per iteration, around the trace, the shade and the march launch
    std::vector<hipEvent_t> pipe_fence;
    bool last_pipeline = false, last_marches = false; int pipe_iters = 0, pipe_timed = 0; unsigned pipe_slots = 0;
    float4 *trace_buf = nullptr; size_t trace_cap = 0;   // rt_trace_*: rays (2 x float4) and hits, reused across calls
    int n_cus = 0;
    unsigned *trace_qc = nullptr;
};
#define RT_PIPE_QN 4096          // ring of per-iteration queue counters
#define RT_PIPE_TIMED 256        // iterations whose trace launch is bracketed by events
#define RT_PIPE_BATCH 4          // iterations launched between two termination checks

template <class T>
static inline int upload(RtScene *s, const T *host, size_t n, const T **dev) {
    void *p = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    HIPCHK(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    if (n) HIPCHK(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<const T *>(p);
    return RT_OK;
}

// (re)allocate a scratch buffer that is only ever used inside one rt_render call
template <class T>
static inline int ensure(RtScene *s, T **buf, size_t *cap, size_t need) {
    if (need <= *cap) return RT_OK;
    if (*buf) { HIPCHK(hipStreamSynchronize(s->stream)); HIPWARN(hipFree(*buf)); *buf = nullptr; *cap = 0; }
    HIPCHK(hipMalloc((void **)buf, need * sizeof(T)));
    *cap = need;
    return RT_OK;
}

// No C++ exception crosses the C boundary: the host builders allocate gigabytes and start worker threads (every group of them is joined while the
// exception unwinds, rt_internal.h ThreadGroup), so bad_alloc / a failed thread start end in a status code, not in std::terminate.
template <class F> static inline int guarded(const char *what, F &&f) {
    try { return f(); }
    catch (const std::bad_alloc &) { return rt::fail(RT_ENOMEM, std::string(what) + ": out of host memory"); }
    catch (const std::exception &e) { return rt::fail(RT_ESTATE, std::string(what) + ": " + e.what()); }
}

// ---- the film gather (rt_film.hip): which of the three kernels a frame takes, and its launch over film rows [row0, row_end) on the scene's stream
struct FilmGather { int which = 0, grx = 0, gry = 0, cols = 0, rows = 0, slot_ncs = 0; size_t slot_lds = 0, col_bytes = 0; };
int film_gather_plan(RtScene *s, const rt::DevFrame &fr, FilmGather &g);
int film_gather_launch(RtScene *s, const rt::DevFrame &fr, const FilmGather &g, const rt::DevFrame *dfr, int row0, int row_end);
