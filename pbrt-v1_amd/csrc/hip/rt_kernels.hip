// rt_kernels.hip -- the render path of libpbrt_hip.so's C ABI (include/pbrt_hip.h): make_frame, the megakernel / queue-pipeline dispatch, rt_render, rt_trace_*.
// (Scene create: rt_scene.hip.  Film kernels and rt_film_*: rt_film.hip.  Shared host declarations: rt_host.h.)
//
// Kernels
//   render_kernel   persistent-thread wavefront renderer: Scene::Render's sample loop (scene.cpp:42-84).
//                   Every lane runs the state machine of rt_integrate.h; all lanes of a wave share ONE
//                   kd-tree traversal loop (rt_traverse.h) whatever kind of ray they carry (camera,
//                   bounce, MIS closest-hit, shadow any-hit); finished lanes refill from a global work
//                   counter with one wave-aggregated atomic.  No MFMA: the work is pointer chasing and
//                   3-vector arithmetic, bounded by HBM/L2 latency and bandwidth, not by dense math.
//   (rt_pipeline.h) pipe_shade_kernel / pipe_trace_kernel: the queue pipeline used for large scenes; rt_trace_closest / rt_trace_any
//                   (unit parity entry points) run caller-supplied rays through the same pipe_trace_kernel.
//   camera_kernel   Sampler + Camera::GenerateRay only.
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (parity with the reference's non-FMA build).
#include "rt_host.h"
#include "rt_weighted.h"

namespace rt {
__global__ void camera_kernel(DevScene sc, DevFrame fr, unsigned long long first, unsigned count, RtRay *out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const unsigned long long n = first + i;
    Lane ln; Ray r;
    setup_sample(sc, fr, ln, n / fr.spp, int(n % fr.spp), r);
    out[i].o[0] = r.o.x; out[i].o[1] = r.o.y; out[i].o[2] = r.o.z;
    out[i].d[0] = r.d.x; out[i].d[1] = r.d.y; out[i].d[2] = r.d.z;
    out[i].mint = r.mint; out[i].maxt = r.maxt;
}

}  // namespace rt

// ------------------------------------------------------------------------------------------ host side
namespace rt {
const char *knob(const char *name) {
#ifdef RT_NO_TUNABLES
    (void)name; return nullptr;
#else
    static const bool on = std::getenv("PBRT_HIP_TUNE") != nullptr;
    return on ? std::getenv(name) : nullptr;
#endif
}
static thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
const char *last_error() { return g_err.c_str(); }
void hip_warn(hipError_t e, const char *what) {
    if (e != hipSuccess) std::fprintf(stderr, "libpbrt_hip: %s failed: %s\n", what, hipGetErrorString(e));
}
}  // namespace rt

// The queue pipeline: alternate a shade kernel (pipe_shade_kernel, or pipe_vertex_kernel for a path without a medium) and pipe_trace_kernel
// until a shade pass enqueues no ray.  The host learns the queue sizes RT_PIPE_BATCH iterations late (page-locked copy + fence event per
// batch), so the GPU never waits for it; the iterations launched after the last productive one find every slot in ST_EXIT and return at once.
static int render_pipeline(RtScene *s, const RtRenderDesc *rd, DevFrame &fr, int vol_levels, int vol_nmax, size_t vol_samp_words) {
    const int integ = rd->integrator;
    // PathIntegrator without a medium: one shade pass per path vertex, all of a vertex's rays in one trace launch (rt_pipe_vertex.h)
    bool by_vertex = integ == RT_INTEGRATOR_PATH && !s->volume.present;
    if (const char *e = knob("PBRT_HIP_PIPE_VERTEX")) by_vertex = by_vertex && std::atoi(e) != 0;
    // ---- pool size: 32 M slots with a medium, 8 M without, unless the frame is smaller or the per-slot scratch would not fit (a fine ray march: 3 floats per step)
    const size_t frame_words = integ != RT_INTEGRATOR_PATH ? size_t(rd->max_depth + 2) * RT_FRAME_WORDS : 0;
    const size_t vol_words = s->volume.present ? size_t(vol_levels) * 8 + 13 + vol_samp_words : 0;
    const size_t slot_bytes = size_t(RT_PIPE_VEC) * 16 + 2 * 16 + 3 * 16 + 4 * 16 + 3 * 4 + (frame_words + vol_words) * 4;
    // (C5, 64 spp with a march per ray: 8 M slots 329 ms, 16 M 311, 32 M 289, 48 M 312, 64 M 326 -- fewer, fuller iterations until the shade passes' state
    // traffic takes over; the by-vertex path frame on the 1 M tree: 8 M 70.6 ms, 16 M 72.1; profiles/r03_c5_knobs.txt)
    unsigned want = s->volume.present ? 1u << 25 : 1u << 23;
    {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t held = size_t(s->pool_cap) * (size_t(RT_PIPE_VEC) * 16 + 9 * 16 + 12) + (s->frames_floats + s->vol_cap) * 4;   // what this scene's pool already holds
        size_t budget = (free_b + held) / 2;                                      // leave half of what is free to the caller (film, other scenes)
        if (const char *e = knob("PBRT_HIP_PIPE_MEM_MB")) budget = size_t(std::max(1, std::atoi(e))) << 20;      // tests: a small budget
        if (slot_bytes * want > budget) want = unsigned(std::max<size_t>(budget / slot_bytes, 2 * RT_BLOCK));
    }
    if (const char *e = knob("PBRT_HIP_PIPE_SLOTS")) want = unsigned(std::max(256, std::atoi(e)));
    unsigned long long tw = fr.total_work ? fr.total_work : 1;
    unsigned n_slots = unsigned(std::min<unsigned long long>(want, tw));
    n_slots = (n_slots + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK;
    if (n_slots >= (1u << 30)) return fail(RT_EINVAL, "rt_render: more than 2^30 pipeline slots");
    const int vec = RT_PIPE_VEC;
    if (n_slots > s->pool_cap) {
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPWARN(hipFree(s->pool.state)); HIPWARN(hipFree(s->pool.ray_o)); HIPWARN(hipFree(s->pool.hit)); HIPWARN(hipFree(s->pool.q_o)); HIPWARN(hipFree(s->pool.q_slot));
        { unsigned *qc = s->pool.q_count; unsigned long long *ww = s->pool.wave_work; s->pool = PipePool{}; s->pool.q_count = qc; s->pool.wave_work = ww; }
        s->pool_cap = 0;
        HIPCHK(hipMalloc((void **)&s->pool.state, size_t(vec) * n_slots * sizeof(float4)));
        HIPCHK(hipMalloc((void **)&s->pool.ray_o, size_t(2) * n_slots * sizeof(float4)));
        HIPCHK(hipMalloc((void **)&s->pool.hit, size_t(3) * n_slots * sizeof(float4)));              // [kind][slot] in the by-vertex form
        HIPCHK(hipMalloc((void **)&s->pool.q_o, size_t(4) * n_slots * sizeof(float4)));
        HIPCHK(hipMalloc((void **)&s->pool.q_slot, size_t(3) * n_slots * sizeof(unsigned)));
        HIPWARN(hipFree(s->pool.wave_work)); s->pool.wave_work = nullptr;
        HIPCHK(hipMalloc((void **)&s->pool.wave_work, size_t(n_slots / 64 + 1) * 2 * sizeof(unsigned long long)));
        s->pool_cap = n_slots;
    }
    if (!s->pool.q_count) HIPCHK(hipMalloc((void **)&s->pool.q_count, size_t(RT_PIPE_QN) * RT_QC_STRIDE * sizeof(unsigned)));
    PipePool pl = s->pool;
    pl.n_slots = n_slots; pl.ray_d = pl.ray_o + n_slots; pl.q_d = pl.q_o + size_t(2) * n_slots; pl.q_march = pl.q_slot + size_t(2) * n_slots;
    if (by_vertex) pl.ray_d = pl.q_o;                                       // directions [3][n_slots] (the compacted ray copies are not used)
    // per-slot scratch of the state machine: recursion frames (whitted / directlighting), volume march state
    if (integ != RT_INTEGRATOR_PATH) {
        int rc = ensure(s, &s->frames, &s->frames_floats, frame_words * n_slots); if (rc) return rc;
    }
    if (s->volume.present) {
        int rc = ensure(s, &s->vol_buf, &s->vol_cap, vol_words * n_slots); if (rc) return rc;
        fr.vol_rays = s->vol_buf; fr.vol_state = s->vol_buf + size_t(vol_levels) * 8 * n_slots;
        fr.vol_samp = fr.vol_state + size_t(13) * n_slots; fr.vol_nmax = vol_nmax;
    }
    fr.frames = s->frames; fr.n_threads = n_slots;
    if (s->pipe_ev.empty()) {
        s->pipe_ev.resize(6 * RT_PIPE_TIMED); s->pipe_fence.resize(RT_PIPE_QN / RT_PIPE_BATCH);
        for (auto &e : s->pipe_ev) HIPCHK(hipEventCreate(&e));
        for (auto &e : s->pipe_fence) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int f = s->counting ? 1 : (s->has_ext ? 2 : 0);
    const PipeShadeFn *st = integ == RT_INTEGRATOR_WHITTED ? g_pipe_shade_whitted : integ == RT_INTEGRATOR_DIRECT ? g_pipe_shade_direct : g_pipe_shade_path;
    const PipeShadeFn shade = by_vertex ? g_pipe_vertex[f] : st[(s->volume.present ? 3 : 0) + f];
    const int tk = (s->accel_kind == RT_ACCEL_GRID ? 4 : 0) + (s->counting ? (s->has_ext ? 1 : 3) : (s->has_ext ? 2 : 0));
    const PipeTraceFn trace = g_pipe_trace[tk];
    const unsigned trace_grid = s->trace_grids[tk];
    // frames with a medium: the march kernel (rt_pipe_march.h) runs the ray marches the shade pass parked
    const bool marches = s->volume.present;
    const int mk = (s->accel_kind == RT_ACCEL_GRID ? 3 : 0) + f;
    const unsigned march_grid = s->march_grids[mk];
    HIPCHK(hipMemcpyAsync(s->dev_frame, &fr, sizeof(DevFrame), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->dev_pool, &pl, sizeof(PipePool), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemsetAsync(s->work_counter, 0, sizeof(unsigned long long), s->stream));
    HIPCHK(hipMemsetAsync(pl.state + n_slots, 0, size_t(n_slots) * sizeof(float4), s->stream));      // control words: every slot in ST_FETCH
    HIPCHK(hipMemsetAsync(pl.wave_work, 0, size_t(n_slots / 64 + 1) * 2 * sizeof(unsigned long long), s->stream));
    HIPCHK(hipEventRecord(s->ev0, s->stream));
    int iter = 0, checked = 0, batch = 0;
    bool done = false;
    const int max_iters = 1 << 20;
    while (!done) {
        for (int k = 0; k < RT_PIPE_BATCH; ++k, ++iter) {
            const unsigned qi = unsigned(iter % RT_PIPE_QN);
            PipeLaunch pk{}; pk.qi = qi; pk.slot_base = 0; pk.q_base = 0;
            HIPCHK(hipMemsetAsync(pl.q_count + size_t(RT_QC_STRIDE) * qi, 0, RT_QC_STRIDE * sizeof(unsigned), s->stream));
            if (iter < RT_PIPE_TIMED) HIPCHK(hipEventRecord(s->pipe_ev[6 * iter + 2], s->stream));
            hipLaunchKernelGGL(shade, dim3(n_slots / RT_BLOCK), dim3(RT_BLOCK), 0, s->stream, (const DevScene *)s->dev_scene,
                               (const DevFrame *)s->dev_frame, (const PipePool *)s->dev_pool, pk);
            if (iter < RT_PIPE_TIMED) HIPCHK(hipEventRecord(s->pipe_ev[6 * iter + 3], s->stream));
            TraceJob job{};
            job.q_o = pl.q_o; job.q_d = pl.q_d; job.q_slot = pl.q_slot; job.q_count = pl.q_count + size_t(RT_QC_STRIDE) * qi; job.hit = pl.hit;
            job.n_slots = n_slots; job.q_base = pk.q_base; job.spill = s->spill; job.n_threads = s->n_threads; job.counters = s->counters;
            if (by_vertex) { job.by_slot = 1; job.q_o = pl.ray_o; job.q_d = pl.ray_d; }
            if (iter < RT_PIPE_TIMED) HIPCHK(hipEventRecord(s->pipe_ev[6 * iter], s->stream));
            hipLaunchKernelGGL(trace, dim3(trace_grid), dim3(RT_BLOCK), 0, s->stream, (const DevScene *)s->dev_scene, job);
            if (iter < RT_PIPE_TIMED) HIPCHK(hipEventRecord(s->pipe_ev[6 * iter + 1], s->stream));
            if (marches) {
                MarchJob mj{};
                mj.q_march = pl.q_march; mj.q_count = pl.q_count + size_t(RT_QC_STRIDE) * qi;
                mj.spill = s->spill; mj.n_threads = s->n_threads; mj.counters = s->counters;
                if (iter < RT_PIPE_TIMED) HIPCHK(hipEventRecord(s->pipe_ev[6 * iter + 4], s->stream));
                hipLaunchKernelGGL(g_pipe_march[mk], dim3(march_grid), dim3(RT_BLOCK), 0, s->stream, (const DevScene *)s->dev_scene, (const DevFrame *)s->dev_frame,
                                   (const PipePool *)s->dev_pool, mj);
                if (iter < RT_PIPE_TIMED) HIPCHK(hipEventRecord(s->pipe_ev[6 * iter + 5], s->stream));
            }
            HIPCHK(hipMemcpyAsync(s->h_qcount + size_t(RT_QC_STRIDE) * qi, pl.q_count + size_t(RT_QC_STRIDE) * qi, (RT_QC_MARCH + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s->pipe_fence[batch % (RT_PIPE_QN / RT_PIPE_BATCH)], s->stream));
        if (batch >= 1) {                                                   // look at the batch before the one just launched
            HIPCHK(hipEventSynchronize(s->pipe_fence[(batch - 1) % (RT_PIPE_QN / RT_PIPE_BATCH)]));
            for (int k = 0; k < RT_PIPE_BATCH; ++k, ++checked) {
                const unsigned *q = s->h_qcount + size_t(RT_QC_STRIDE) * (checked % RT_PIPE_QN);
                if (q[0] + q[RT_QC_ANY] + q[RT_QC_MARCH] == 0) { done = true; break; }
            }
        }
        ++batch;
        if (iter > max_iters) return fail(RT_ESTATE, "rt_render: the queue pipeline did not terminate");
    }
    s->last_marches = marches;
    s->pipe_slots = n_slots; s->pipe_iters = checked + 1; s->pipe_timed = std::min(s->pipe_iters, RT_PIPE_TIMED);
    HIPCHK(hipEventRecord(s->ev1, s->stream));
    s->last_pipeline = true;
    return RT_OK;
}

extern "C" {


const char *rt_last_error(void) { return rt::last_error(); }

int rt_device_count(int *count) {
    if (!count) return fail(RT_EINVAL, "rt_device_count: null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RT_EDEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n; return RT_OK;
}
// Build the per-frame device descriptor: film geometry + the Sample layout the integrators request
// (Sample::Sample sampling.cpp:41-70; RequestSamples of directlighting.cpp:39-66, path.cpp:47-57,
// emission.cpp:42-46 / single.cpp:43-47; LatinHypercube draw counts sampling.cpp:98-113).
static int make_frame(RtScene *s, const RtRenderDesc *rd, DevFrame &fr, bool need_film) {
    auto round_up_pow2 = [](unsigned v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; };   // pbrt.h:590-598
    if (rd->sampler < RT_SAMPLER_STRATIFIED || rd->sampler > RT_SAMPLER_RANDOM) return fail(RT_EINVAL, "unknown sampler");
    if (rd->sampler == RT_SAMPLER_LOWDISCREPANCY) { if (rd->pixel_samples < 1) return fail(RT_EINVAL, "bad pixelsamples"); }
    else if (rd->x_samples < 1 || rd->y_samples < 1) return fail(RT_EINVAL, "bad xsamples/ysamples");
    std::memset(&fr, 0, sizeof fr);
    fr.integrator = rd->integrator; fr.max_depth = rd->max_depth; fr.strategy = rd->strategy;
    fr.volume_integrator = rd->volume_integrator; fr.step_size = rd->step_size;
    fr.sampler = rd->sampler; fr.xs = rd->x_samples; fr.ys = rd->y_samples; fr.jitter = rd->jitter;
    fr.spp = rd->sampler == RT_SAMPLER_LOWDISCREPANCY ? int(round_up_pow2(unsigned(rd->pixel_samples))) : rd->x_samples * rd->y_samples;
    fr.seed = rd->seed;
    fr.x_pixel_start = rd->x_pixel_start; fr.y_pixel_start = rd->y_pixel_start;
    fr.x_pixel_count = rd->x_pixel_count; fr.y_pixel_count = rd->y_pixel_count;
    fr.x_start = rd->x_start; fr.x_end = rd->x_end; fr.y_start = rd->y_start; fr.y_end = rd->y_end;
    if (fr.x_end <= fr.x_start || fr.y_end <= fr.y_start) return fail(RT_EINVAL, "empty sample extent");
    fr.fxw = rd->filter_x_width; fr.fyw = rd->filter_y_width;
    fr.inv_fxw = 1.f / fr.fxw; fr.inv_fyw = 1.f / fr.fyw;
    fr.filter_table = s->filter_dev; fr.accum = s->accum;
    fr.shard_index = rd->shard_index; fr.shard_count = rd->shard_count < 1 ? 1 : rd->shard_count;
    fr.tile_pixels = rd->tile_pixels < 1 ? 1 : rd->tile_pixels;
    if (fr.shard_index < 0 || fr.shard_index >= fr.shard_count) return fail(RT_EINVAL, "bad shard index");
    fr.total_pixels = (unsigned long long)(fr.x_end - fr.x_start) * (unsigned long long)(fr.y_end - fr.y_start);
    if (fr.total_pixels * fr.spp > 0xFFFFFFFFull) return fail(RT_EINVAL, "more than 2^32 camera samples per frame");
    unsigned long long n_tiles = 0;
    if (rd->tile_pixels < 0) {                               // 2-D tiles: width in the low 16 bits of -tile_pixels, height above (pbrt_hip.h)
        const int tw = (-rd->tile_pixels) & 0xffff, th = (-rd->tile_pixels) >> 16;
        if (tw < 1 || th < 1 || tw > 4096 || th > 4096) return fail(RT_EINVAL, "bad 2-D tile size");
        fr.tile_w = tw; fr.tile_h = th; fr.tile_pixels = tw * th;
        fr.tiles_x = (fr.x_end - fr.x_start + tw - 1) / tw;
        n_tiles = (unsigned long long)fr.tiles_x * (unsigned long long)((fr.y_end - fr.y_start + th - 1) / th);
        if (n_tiles * fr.tile_pixels * fr.spp > 0xFFFFFFFFull) return fail(RT_EINVAL, "more than 2^32 camera samples per frame (2-D tiles, border tiles included)");
    } else n_tiles = (fr.total_pixels + fr.tile_pixels - 1) / fr.tile_pixels;
    const unsigned long long my_tiles = n_tiles > (unsigned long long)fr.shard_index
        ? (n_tiles - fr.shard_index + fr.shard_count - 1) / fr.shard_count : 0;
    fr.total_work = my_tiles * fr.tile_pixels * fr.spp;

    // sample layout
    std::vector<int> n1, n2;
    const int nl = int(s->dev.n_lights);
    if (rd->integrator == RT_INTEGRATOR_DIRECT && rd->strategy == RT_STRATEGY_ALL) {
        std::vector<DevLight> lights(nl);
        if (nl) HIPCHK(hipMemcpy(lights.data(), s->dev.lights, nl * sizeof(DevLight), hipMemcpyDeviceToHost));
        for (int i = 0; i < nl; ++i) { int ns = lights[i].n_samples; if (rd->sampler == RT_SAMPLER_LOWDISCREPANCY) ns = int(round_up_pow2(unsigned(ns)));   // Sampler::RoundSize
            if (ns > 65535) return fail(RT_EINVAL, "rt_render: more than 65535 samples per light (the sample table holds counts in 16 bits)");
            n2.push_back(ns); n2.push_back(ns); n1.push_back(ns); }
    } else if (rd->integrator == RT_INTEGRATOR_DIRECT) { n2 = {1, 1}; n1 = {1, 1}; }
    else if (rd->integrator == RT_INTEGRATOR_PATH) { n1.assign(9, 1); n2.assign(9, 1); }
    else if (rd->integrator != RT_INTEGRATOR_WHITTED) return fail(RT_EINVAL, "unknown integrator");
    n1.push_back(1); n1.push_back(1);                   // the volume integrator's tau / scatter samples
    // The requests in the reference's order (every 1-D request, then every 2-D one: Sample::Sample sampling.cpp:41-70).  DirectLighting "all" asks for
    // 2 x 2-D and 1 x 1-D per light WITHOUT bound (directlighting.cpp:39-66): its per-light requests go to a table in HBM (DevFrame::light_dims, three
    // records per light, indexed by the lane's light cursor); the frame descriptor itself carries the bounded rest.
    const bool all_lights = rd->integrator == RT_INTEGRATOR_DIRECT && rd->strategy == RT_STRATEGY_ALL;
    std::vector<DimReq> d1(n1.size()), d2(n2.size());
    unsigned c = 0;
    const unsigned P = unsigned(fr.spp);
    if (rd->sampler == RT_SAMPLER_STRATIFIED) {             // LatinHypercube: n*d floats then n*d shuffles per request
        for (size_t i = 0; i < n1.size(); ++i) { d1[i] = DimReq{c, c + unsigned(n1[i]), (unsigned short)n1[i], 1}; c += 2u * n1[i]; }
        for (size_t i = 0; i < n2.size(); ++i) { d2[i] = DimReq{c, c + 2u * n2[i], (unsigned short)n2[i], 2}; c += 4u * n2[i]; }
        fr.lhs_total = c;
        fr.pixgen_draws = fr.jitter ? 7u * P : 2u * P;      // stratified.cpp:99-117
    } else if (rd->sampler == RT_SAMPLER_RANDOM) {          // one float per value (random.cpp:107-112)
        for (size_t i = 0; i < n1.size(); ++i) { d1[i] = DimReq{c, 0, (unsigned short)n1[i], 1}; c += unsigned(n1[i]); }
        for (size_t i = 0; i < n2.size(); ++i) { d2[i] = DimReq{c, 0, (unsigned short)n2[i], 2}; c += 2u * n2[i]; }
        fr.lhs_total = c;
        fr.pixgen_draws = 5u * P;                           // random.cpp:88-92
    } else {                                                // per-pixel tables (lowdiscrepancy.cpp:93-104, sampling.h:152-174)
        c = (2 + 2 * P) + (2 + 2 * P) + (1 + 2 * P);        // image, lens, time blocks
        for (size_t i = 0; i < n1.size(); ++i) { d1[i] = DimReq{c, 0, (unsigned short)n1[i], 1}; c += 1u + unsigned(n1[i]) * P + P; }
        for (size_t i = 0; i < n2.size(); ++i) { d2[i] = DimReq{c, 0, (unsigned short)n2[i], 2}; c += 2u + unsigned(n2[i]) * P + P; }
        fr.lhs_total = 0;
        fr.pixgen_draws = c;
    }
    {   // the draws of one camera sample are addressed by a 32-bit counter: the tables above must fit well inside it
        unsigned long long draws = 0;
        for (int n : n1) draws += 2ull * unsigned(n) * (P + 1);
        for (int n : n2) draws += 4ull * unsigned(n) * (P + 1);
        if (draws > 0x3fffffffull) return fail(RT_EINVAL, "rt_render: the lights' sample requests need more than 2^30 random numbers per camera sample");
    }
    fr.light_dims = nullptr;
    fr.dims_max_n = 0;
    for (const DimReq &r : d2) fr.dims_max_n = std::max(fr.dims_max_n, int(r.n));
    if (all_lights) {
        std::vector<DimReq> tab(size_t(nl) * 3 + 1, DimReq{0, 0, 0, 0});
        for (int i = 0; i < nl; ++i) { tab[3 * size_t(i)] = d2[2 * size_t(i)]; tab[3 * size_t(i) + 1] = d2[2 * size_t(i) + 1]; tab[3 * size_t(i) + 2] = d1[size_t(i)]; }
        std::vector<DimReq> &up = s->light_dims_host;          // what the device holds: uploaded again only when a frame asks for other requests
        if (!s->light_dims || up.size() != tab.size() || std::memcmp(up.data(), tab.data(), tab.size() * sizeof(DimReq)) != 0) {
            int rc = ensure(s, &s->light_dims, &s->light_dims_cap, tab.size()); if (rc) return rc;
            up.swap(tab);
            // rt_render is asynchronous: a frame queued earlier may still be reading the table, and the source is pageable host memory that the next make_frame
            // may swap away -- wait for the stream, then copy synchronously (36 bytes per light; only when a frame asks for other requests than the last one)
            HIPCHK(hipStreamSynchronize(s->stream));
            HIPCHK(hipMemcpy(s->light_dims, up.data(), up.size() * sizeof(DimReq), hipMemcpyHostToDevice));
        }
        fr.light_dims = s->light_dims;
        d1.erase(d1.begin(), d1.begin() + nl);              // what stays in the descriptor: the volume integrator's two 1-D requests
        d2.clear();
    }
    if (d1.size() > RT_MAX_DIM_REQ || d2.size() > RT_MAX_DIM_REQ) return fail(RT_ESTATE, "rt_render: sample table overflow");      // (cannot happen: 11 / 9 for the path integrator)
    fr.n1d = int(d1.size()); fr.n2d = int(d2.size());
    std::memset(fr.one_d, 0, sizeof fr.one_d); std::memset(fr.two_d, 0, sizeof fr.two_d);
    for (size_t i = 0; i < d1.size(); ++i) fr.one_d[i] = d1[i];
    for (size_t i = 0; i < d2.size(); ++i) fr.two_d[i] = d2[i];
    // traversal scheduling knobs (performance only; results and counters do not depend on them)
    {
        const size_t nn = s->tree.nodes.size();
        if (s->per_leaf < 0.0) {                              // once per scene: a pass over 36 M nodes costs 12 ms, not something to pay per frame
            size_t leaves = 0, refs = 0;
            if (s->accel_kind == RT_ACCEL_KDTREE) for (const Node &n : s->tree.nodes) if ((n.x & 3u) == 3u && (n.x >> 2)) { ++leaves; refs += n.x >> 2; }
            s->per_leaf = leaves ? double(refs) / double(leaves) : 1.0;
        }
        const double per_leaf = s->per_leaf;
        const bool tiny = nn <= 4096 && per_leaf >= 1.5;       // few fat leaves: triangle tests dominate -> lock-step rounds
        const bool tiny_path = tiny && rd->integrator == RT_INTEGRATOR_PATH;
        fr.trav_mode = (tiny && !tiny_path) ? 3 : 2;           // tiny trees, Whitted / DirectLighting: lock-step rounds with pooled leaf tests; else batched
                                                               // rounds (measured best on 100k-1M triangle soups and, since the spill-free round-3 build,
                                                               // for the path integrator on tiny trees too: profiles/r03_c2_knobs.txt)
        // long divergent rays: let finished lanes refill early.  Tiny scenes: only with phase gating (path integrator), where 8
        // measured +1.5 % (16: -13 %); Whitted / DirectLighting on Cornell lose 10 % with any early exit
        fr.exit_thresh = tiny ? (rd->integrator == RT_INTEGRATOR_PATH ? 8 : 0) : 32;
        // round 5, the path integrator by vertex (a lane comes back for shading once per vertex, not once per ray): 16 on tiny trees (C2's kernel 49.8 -> 48.5 ms;
        // 24: 48.7, 32: 50.1), 32-40 alike on the 1 M-triangle frames (profiles/r05_by_vertex_scan.txt)
        if (RT_MEGA_BYV && tiny && rd->integrator == RT_INTEGRATOR_PATH && !s->volume.present && !s->has_ext) fr.exit_thresh = 16;
        fr.high_occupancy = (tiny && !tiny_path) ? 0 : 1;      // C2 (round 3): 4-wave flavour + batched rounds + exit threshold 8 = 54.0 ms, natural allocation + pooled lock-step 56.5
        if (const char *e = knob("PBRT_HIP_HIGH_OCC")) fr.high_occupancy = std::atoi(e);
        fr.leaf_min = tiny ? 8 : RT_TRACE_LEAF_MIN;          // C2: 50.8 ms at 8, 54.0 at 24 (few fat leaves: waiting for a fuller batch only idles lanes)
        if (const char *e = knob("PBRT_HIP_LEAF_MIN")) fr.leaf_min = std::max(1, std::atoi(e));
        if (const char *e = knob("PBRT_HIP_TRAV_MODE")) fr.trav_mode = std::atoi(e);
        // XCD bands of the work list (rt_render_kernel.h): frames whose rays stay coherent (Whitted / DirectLighting: camera, shadow and specular rays)
        // gain from each private L2 holding one band's lines -- C3 13.37 -> 12.62 ms; path frames, whose rays scatter after the first bounce, lose
        // 5 % (1 M path 62.2 -> 65.5 ms, C4 64.1 -> 67.5; profiles/r04_xcd_bands_scan.txt)
        fr.xcd_bands = rd->integrator != RT_INTEGRATOR_PATH ? 1 : 0;
        if (const char *e = knob("PBRT_HIP_XCD_BANDS")) fr.xcd_bands = std::atoi(e) != 0;
        if (const char *e = knob("PBRT_HIP_EXIT_THRESH")) fr.exit_thresh = std::atoi(e);
        fr.dbg_x = fr.dbg_y = -1000000;
        if (const char *e = knob("PBRT_HIP_DEBUG_PIXEL")) std::sscanf(e, "%d,%d", &fr.dbg_x, &fr.dbg_y);
        fr.phase_sync = tiny ? 1 : 0;                          // C2: 63.6 vs 82.4 ms; 100k/1M soups (early-exit rounds): 8 % slower
        if (const char *e = knob("PBRT_HIP_PHASE_SYNC")) fr.phase_sync = std::atoi(e);
        // large trees (traversal bound by memory latency): the queue pipeline of rt_pipeline.h; tiny cache-resident ones: the megakernel
        // (measured on the 1 M-triangle frames, 1x MI355X: the pipeline's trace kernel is faster than the megakernel's traversal, but its
        // state traffic and sparse last iterations cost more than that gains, except where shading suspends often: volume marching)
        // round 3: a path without a medium can take the by-vertex form (rt_pipe_vertex.h: 1 M-triangle frame 69.9 ms); since the kernels are
        // built without the SLP vectorizer the 4-wave megakernel no longer spills and is as fast there (69.3 ms) and faster on C4's
        // material mix (71.3 vs 76.3 ms), so it stays the default for frames without a medium
        fr.pipeline = (!tiny && s->volume.present) ? 1 : 0;
        if (const char *e = knob("PBRT_HIP_PIPELINE")) fr.pipeline = std::atoi(e) != 0;
        if (fr.max_depth > 250 || fr.max_depth < 0) fr.pipeline = 0;     // the slot's control word holds depth in 8 bits
        if (fr.dims_max_n >= 65535) fr.pipeline = 0;                     // ... and the light / sample cursors in 16 bits each
        if (s->dev.n_lights >= 65535u) fr.pipeline = 0;
        if (rd->integrator == RT_INTEGRATOR_DIRECT && rd->strategy == RT_STRATEGY_WEIGHTED) fr.pipeline = 0;    // three megakernel passes (rt_weighted.h)
        if (fr.shard_count == 1 && !fr.pipeline) {
            // One shard: the tiles partition nothing, and the megakernel then renders the sample extent in scanline order.  2-D tiles pad the extent to
            // whole tiles and every dropped padding item idles a lane for about a ray's time: 64 x 64 tiles cost C3 5 % of its frame
            // (profiles/r03_work_order.txt).  The queue pipeline keeps the caller's tiles: with millions of paths in flight their compactness is
            // worth more (C5: L2 misses 5.3 G per frame in 64 x 64 tiles, 6.3 G in scanline order; 328 vs 332 ms).
            fr.tile_w = fr.tile_h = fr.tiles_x = 0; fr.tile_pixels = 1;
            fr.total_work = fr.total_pixels * fr.spp;
            // The counter hands the scanline-addressed samples out in 32 x 32-pixel blocks (clipped at the extent's edges: no padding, nothing dropped):
            // C3's kernel 12.68 -> 12.18 ms on top of the XCD bands, the path frames and C2 unchanged (profiles/r04_mega_tile_scan.txt).
            fr.mega_tile = 32;
            if (const char *e = knob("PBRT_HIP_MEGA_TILE")) fr.mega_tile = std::max(0, std::atoi(e));
            if (fr.mega_tile > 4096 || fr.total_work > 0xffffffffull) fr.mega_tile = 0;
        }
        if (fr.trav_mode != 3 || fr.high_occupancy) fr.trav_mode = 2;  // (the high-occupancy kernels carry no pooled-leaf scratch)
    }
    fr.work_counter = s->work_counter; fr.counters = s->counters; fr.spill = s->spill; fr.n_threads = s->n_threads;
    fr.frames = s->frames;
    if (need_film && !s->accum) return fail(RT_ESTATE, "rt_render: no film bound (call rt_film_bind first)");
    if (need_film && (fr.x_pixel_count != s->film_w || fr.y_pixel_count != s->film_h))
        return fail(RT_EINVAL, "rt_render: film size does not match the bound film");
    return RT_OK;
}

int rt_camera_rays(RtScene *s, const RtRenderDesc *rd, uint64_t first, uint32_t count, RtRay *rays_out) {
    if (!s || !rd || !rays_out) return fail(RT_EINVAL, "null argument");
    HIPCHK(hipSetDevice(s->device));
    DevFrame fr; int rc = make_frame(s, rd, fr, false); if (rc) return rc;
    RtRay *dev = nullptr;
    HIPCHK(hipMalloc((void **)&dev, size_t(count ? count : 1) * sizeof(RtRay)));
    hipLaunchKernelGGL(camera_kernel, dim3((count + 255) / 256), dim3(256), 0, s->stream, s->dev, fr,
                       (unsigned long long)first, count, dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(rays_out, dev, size_t(count) * sizeof(RtRay), hipMemcpyDeviceToHost));
    HIPWARN(hipFree(dev));
    return RT_OK;
}

// Scene::Intersect / IntersectP for caller-supplied rays: the rays form one queue of the pipeline's trace kernel (counting
// twin with the quadric code), so the unit parity tests exercise the very kernel the large-scene renders spend their time in.
static int trace_common(RtScene *s, const RtRay *rays, uint32_t n, int any, RtHit *hits, uint8_t *occ) {
    if (!s || !rays || (!hits && !occ)) return fail(RT_EINVAL, "null argument");
    HIPCHK(hipSetDevice(s->device));
    if (n == 0) return RT_OK;
    if (size_t(n) > s->trace_cap) {
        if (s->trace_buf) { HIPCHK(hipStreamSynchronize(s->stream)); HIPWARN(hipFree(s->trace_buf)); s->trace_buf = nullptr; s->trace_cap = 0; }
        HIPCHK(hipMalloc((void **)&s->trace_buf, size_t(n) * 3 * sizeof(float4))); s->trace_cap = n;
    }
    std::vector<float4> host(size_t(n) * 2);
    for (uint32_t i = 0; i < n; ++i) {
        host[i] = make_float4(rays[i].o[0], rays[i].o[1], rays[i].o[2], rays[i].mint);
        host[size_t(n) + i] = make_float4(rays[i].d[0], rays[i].d[1], rays[i].d[2], rays[i].maxt);
    }
    unsigned qc[RT_QC_STRIDE] = {0};
    qc[0] = any ? 0u : n; qc[RT_QC_ANY] = any ? n : 0u;
    HIPCHK(hipMemcpyAsync(s->trace_buf, host.data(), host.size() * sizeof(float4), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->trace_qc, qc, sizeof qc, hipMemcpyHostToDevice, s->stream));
    TraceJob job{};
    job.q_o = s->trace_buf; job.q_d = s->trace_buf + n; job.q_slot = nullptr; job.q_count = s->trace_qc; job.hit = s->trace_buf + 2 * size_t(n);
    job.n_slots = 0; job.spill = s->spill; job.n_threads = s->n_threads; job.counters = s->counters;
    const int k = (s->accel_kind == RT_ACCEL_GRID ? 4 : 0) + (s->has_ext ? 1 : 3);
    HIPWARN(hipEventRecord(s->ev0, s->stream));
    hipLaunchKernelGGL(g_pipe_trace[k], dim3(s->trace_grids[k]), dim3(RT_BLOCK), 0, s->stream, (const DevScene *)s->dev_scene, job);
    HIPWARN(hipEventRecord(s->ev1, s->stream)); HIPWARN(hipEventRecord(s->ev2, s->stream)); s->have_timing = true; s->last_pipeline = false;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host.data(), s->trace_buf + 2 * size_t(n), size_t(n) * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (uint32_t i = 0; i < n; ++i) {
        int prim; std::memcpy(&prim, &host[i].x, 4);
        if (any) occ[i] = prim >= 0 ? 1 : 0;
        else { hits[i].prim = prim; hits[i].t = host[i].y; hits[i].b1 = host[i].z; hits[i].b2 = host[i].w; }
    }
    return RT_OK;
}
int rt_trace_closest(RtScene *s, const RtRay *rays, uint32_t n, RtHit *hits_out) { return trace_common(s, rays, n, 0, hits_out, nullptr); }
int rt_trace_any(RtScene *s, const RtRay *rays, uint32_t n, uint8_t *occluded_out) { return trace_common(s, rays, n, 1, nullptr, occluded_out); }
int rt_render(RtScene *s, const RtRenderDesc *rd) {
    if (!s || !rd) return fail(RT_EINVAL, "null argument");
    HIPCHK(hipSetDevice(s->device));
    // ---- validation first: a failing call launches nothing and leaves the film and the sample buffer untouched
    if (rd->integrator < RT_INTEGRATOR_WHITTED || rd->integrator > RT_INTEGRATOR_PATH) return fail(RT_EINVAL, "unknown integrator");
    if (rd->max_depth < 0) return fail(RT_EINVAL, "rt_render: negative maxdepth");
    if (!rd->filter_table) return fail(RT_EINVAL, "rt_render: no filter table");
    DevFrame fr; int rc = make_frame(s, rd, fr, true); if (rc) return rc;
    const bool weighted = rd->integrator == RT_INTEGRATOR_DIRECT && rd->strategy == RT_STRATEGY_WEIGHTED;
    if (rd->integrator == RT_INTEGRATOR_DIRECT && (rd->strategy < RT_STRATEGY_ALL || rd->strategy > RT_STRATEGY_WEIGHTED)) return fail(RT_EINVAL, "rt_render: unknown direct lighting strategy");
    if (weighted) {                                           // WeightedSampleOneLight: rt_weighted.h
        if (fr.shard_count != 1) return fail(RT_EINVAL, "rt_render: strategy \"weighted\" is a recurrence over the whole frame in the sampler's order (transport.cpp:71-122): one shard only");
        // (lights of mixed RNG use -- an emitter of several triangles draws its triangle, every other light draws nothing -- run the general form of the
        // survey and the recurrence: DevFrame::wt_mixed)
        // With a medium EstimateDirect draws one RandomFloat per UNOCCLUDED shadow / MIS ray (Scene::Transmittance), so the draws of an estimate depend on
        // the light AND on occlusion: the survey pass (every light, the counter left where the LAST light's estimate ends) and the frame pass (the chosen
        // light only) then reach the sample's next shading point with different counters, and a light that draws (ShapeSet::Sample's triangle pick) is fed
        // a different random number than the one its surveyed luminance came from.  Delta lights never draw: their Ld does not depend on it.
        if (s->volume.present && s->light_draws != 0) return fail(RT_EINVAL, "rt_render: strategy \"weighted\" in a participating medium needs lights whose estimates draw no random numbers "
                                                                            "(point / spot / distant / single-triangle / quadric lights): an emitter of several triangles draws its triangle, and the medium makes "
                                                                            "the draw's position in the stream depend on occlusion");
        if (s->dev.n_lights > 2048u) return fail(RT_EINVAL, "rt_render: strategy \"weighted\" holds the per-light tables of its recurrence in LDS: at most 2048 lights");
        if (fr.total_work >= 0xffffffffull) return fail(RT_EINVAL, "rt_render: strategy \"weighted\": more than 2^32 - 2 camera samples in the frame");
    }
    const bool skip_film = knob("PBRT_HIP_DEBUG_NOFILM") != nullptr;   // perf experiments only
    FilmGather fg; rc = film_gather_plan(s, fr, fg); if (rc) return rc;
    int vol_levels = 0, vol_nmax = 0; size_t vol_samp_words = 0;
    if (s->volume.present) {
        if (!(rd->step_size > 0.f)) return fail(RT_EINVAL, "rt_render: volume integrator stepsize must be positive");
        if (rd->volume_integrator != RT_VOLUME_EMISSION && rd->volume_integrator != RT_VOLUME_SINGLE) return fail(RT_EINVAL, "rt_render: unknown volume integrator");
        const float ex = s->volume.p1[0] - s->volume.p0[0], ey = s->volume.p1[1] - s->volume.p0[1], ez = s->volume.p1[2] - s->volume.p0[2];
        const double diag = std::sqrt(double(ex) * ex + double(ey) * ey + double(ez) * ez);
        const double nsteps = std::ceil(diag / rd->step_size) + 2;
        if (nsteps > 65536) return fail(RT_EINVAL, "rt_render: stepsize too small for the medium (more than 65536 march steps)");
        vol_nmax = int(nsteps);
        vol_levels = (rd->integrator == RT_INTEGRATOR_PATH) ? 1 : rd->max_depth + 2;
        vol_samp_words = rd->volume_integrator == RT_VOLUME_SINGLE ? size_t(3) * vol_nmax : 0;
    }
    // ---- scratch
    {
        size_t cap = s->samples_cap;
        const size_t local_pixels = size_t(fr.total_work / fr.spp);                     // sample_slot(): whole 64-pixel chunks
        const size_t need = ((local_pixels + 63) / 64 * 64 + 64) * size_t(fr.spp) * 2;
        rc = ensure(s, &s->samples, &cap, need); if (rc) return rc;
        // A fresh buffer is zeroed once: the film gathers weigh the records of lanes without a column of their own (film_march_kernel: record 0)
        // and of dropped border items with +0, which is only a zero if the record is finite -- from here on sample_write keeps it so; a rank
        // that owns no tile at all writes nothing, ever.  For dropped items rt_samples_read returns zeros or what an earlier frame left there.
        if (cap != s->samples_cap) HIPCHK(hipMemsetAsync(s->samples, 0, cap * sizeof(float4), s->stream));
        s->samples_cap = cap;
    }
    fr.samples = s->samples; s->samples_last = fr.total_work; s->samples_spp = fr.spp;
    HIPCHK(hipMemcpyAsync(s->filter_dev, rd->filter_table, 256 * sizeof(float), hipMemcpyHostToDevice, s->stream));
    { hipError_t pre = hipGetLastError(); if (pre != hipSuccess) return fail(RT_EDEVICE, std::string("pending HIP error before launch: ") + hipGetErrorString(pre)); }
    if (fr.pipeline) {
        rc = render_pipeline(s, rd, fr, vol_levels, vol_nmax, vol_samp_words); if (rc) return rc;
        s->last_weighted = false;
    } else {
        if (rd->integrator != RT_INTEGRATOR_PATH) {           // recursion frames for whitted / directlighting
            rc = ensure(s, &s->frames, &s->frames_floats, size_t(rd->max_depth + 2) * RT_FRAME_WORDS * s->n_threads); if (rc) return rc;
            fr.frames = s->frames;
        }
        if (s->volume.present) {
            rc = ensure(s, &s->vol_buf, &s->vol_cap, (size_t(vol_levels) * 8 + 13 + vol_samp_words) * s->n_threads); if (rc) return rc;
            fr.vol_rays = s->vol_buf; fr.vol_state = s->vol_buf + size_t(vol_levels) * 8 * s->n_threads;
            fr.vol_samp = fr.vol_state + size_t(13) * s->n_threads; fr.vol_nmax = vol_nmax;
        }
        int variant = (((s->volume.present ? 1 : 0) * 2 + (s->accel_kind == RT_ACCEL_GRID ? 1 : 0)) * 2 + (s->counting ? 1 : 0)) * 3 + rd->integrator;
        if (fr.high_occupancy && !s->counting) variant = 24 + ((s->volume.present ? 1 : 0) * 2 + (s->accel_kind == RT_ACCEL_GRID ? 1 : 0)) * 3 + rd->integrator;
        if (s->has_ext && !s->counting) variant = 36 + ((s->volume.present ? 1 : 0) * 2 + (s->accel_kind == RT_ACCEL_GRID ? 1 : 0)) * 3 + rd->integrator;
        if (s->grids[variant] == 0) return fail(RT_ESTATE, "render kernel variant has no resident grid");
        fr.weighted_phase = 0; fr.wt_base = nullptr; fr.wt_rec = nullptr; fr.wt_pick = nullptr; fr.wt_mixed = 0; fr.wt_nd = 0; fr.wt_recbase = nullptr;
        if (weighted) {
            // rt_weighted.h: count -> scan -> survey -> recurrence -> the frame.  ev0 .. ev1 bracket all of it (kernel_ms of a weighted frame is the five together)
            const int wk = ((s->volume.present ? 1 : 0) * 2 + (s->accel_kind == RT_ACCEL_GRID ? 1 : 0)) * 2 + (s->counting ? 1 : 0);
            const int nL = int(s->dev.n_lights), R = 1 + 2 * nL;
            if (!s->wt_total) { HIPCHK(hipHostMalloc((void **)&s->wt_total, sizeof(unsigned long long), hipHostMallocDefault)); HIPCHK(hipMalloc((void **)&s->wt_sums, RT_WSCAN_BLOCKS * sizeof(unsigned long long))); for (hipEvent_t &e : s->wt_ev) HIPCHK(hipEventCreate(&e)); }
            rc = ensure(s, &s->wt_base, &s->wt_base_cap, size_t(fr.total_work) + 1); if (rc) return rc;
            fr.wt_base = s->wt_base;
            auto pass = [&](int phase) -> int {
                fr.weighted_phase = phase;
                HIPCHK(hipMemcpyAsync(s->dev_frame, &fr, sizeof(DevFrame), hipMemcpyHostToDevice, s->stream));
                HIPCHK(hipMemsetAsync(s->work_counter, 0, 64 * sizeof(unsigned long long), s->stream));
                hipLaunchKernelGGL(g_render_kernels_weighted[wk], dim3(s->wgrids[wk]), dim3(RT_BLOCK), 0, s->stream, (const DevScene *)s->dev_scene, (const DevFrame *)s->dev_frame);
                HIPCHK(hipGetLastError());
                return RT_OK;
            };
            HIPCHK(hipEventRecord(s->ev0, s->stream));
            if ((rc = pass(1))) return rc;
            HIPCHK(hipEventRecord(s->wt_ev[0], s->stream));
            hipLaunchKernelGGL(weighted_scan_sums_kernel, dim3(RT_WSCAN_BLOCKS), dim3(RT_WSCAN_THREADS), 0, s->stream, (const unsigned *)s->wt_base, (unsigned long long)fr.total_work, s->wt_sums);
            hipLaunchKernelGGL(weighted_scan_top_kernel, dim3(1), dim3(RT_WSCAN_THREADS), 0, s->stream, s->wt_base, (unsigned long long)fr.total_work, s->wt_sums, s->wt_total);
            hipLaunchKernelGGL(weighted_scan_apply_kernel, dim3(RT_WSCAN_BLOCKS), dim3(RT_WSCAN_THREADS), 0, s->stream, s->wt_base, (unsigned long long)fr.total_work, (const unsigned long long *)s->wt_sums);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(s->wt_ev[1], s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
            const unsigned long long n_points = *s->wt_total;
            // The one place where rt_render can fail AFTER it has launched work (the number of shading points is only known once the count pass has run;
            // pbrt_hip.h says so): the film and the sample buffer are untouched (the count pass writes neither), the scene is left without a last frame.
            auto late_fail = [&](int code, const char *msg) { s->last_weighted = false; s->last_pipeline = false; s->have_timing = false; s->wt_points = 0; return fail(code, msg); };
            if (n_points >= 0xffffffffull) return late_fail(RT_EINVAL, "rt_render: strategy \"weighted\": more than 2^32 - 2 shading points in the frame");
            s->wt_points = n_points;
            const bool mixed = s->light_draws < 0;
            const int nD = int(s->n_drawing_lights);
            size_t rec_floats = size_t(n_points) * size_t(R);
            fr.wt_mixed = mixed ? 1 : 0; fr.wt_nd = unsigned(nD); fr.wt_recbase = nullptr;
            if (mixed) {
                // A sample with P shading points keeps P * A + nD * P * (P + 1) floats: quadratic in P, and sum P (P + 1) >= W * Pavg * (Pavg + 1) for W samples of
                // Pavg points on average (convexity) -- a frame that cannot fit even by that lower bound is refused here, before the size scan is launched
                {
                    const double W = double(fr.total_work ? fr.total_work : 1), Pavg = double(n_points) / W;
                    const double at_least = double(n_points) * double(1 + 2 * (nL - nD)) + double(nD) * W * Pavg * (Pavg + 1.0);
                    if (at_least >= 4294967295.0) return late_fail(RT_EINVAL, "rt_render: strategy \"weighted\" with lights of mixed RNG use: the survey's records exceed 2^32 floats (quadratic in a sample's shading points)");
                }
                // where every sample's records start: per-sample sizes from the point counts (P * A + nD * P * (P + 1)), scanned like the counts were
                rc = ensure(s, &s->wt_recbase, &s->wt_recbase_cap, size_t(fr.total_work) + 1); if (rc) return late_fail(rc, rt_last_error());
                const unsigned long long nw = fr.total_work;
                if (nw) hipLaunchKernelGGL(weighted_sizes_kernel, dim3(unsigned((nw + 255) / 256)), dim3(256), 0, s->stream, (const unsigned *)s->wt_base, nw, unsigned(1 + 2 * (nL - nD)), unsigned(nD), s->wt_recbase);
                hipLaunchKernelGGL(weighted_scan_sums_kernel, dim3(RT_WSCAN_BLOCKS), dim3(RT_WSCAN_THREADS), 0, s->stream, (const unsigned *)s->wt_recbase, nw, s->wt_sums);
                hipLaunchKernelGGL(weighted_scan_top_kernel, dim3(1), dim3(RT_WSCAN_THREADS), 0, s->stream, s->wt_recbase, nw, s->wt_sums, s->wt_total);
                hipLaunchKernelGGL(weighted_scan_apply_kernel, dim3(RT_WSCAN_BLOCKS), dim3(RT_WSCAN_THREADS), 0, s->stream, s->wt_recbase, nw, (const unsigned long long *)s->wt_sums);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(s->stream));
                if (*s->wt_total >= 0xffffffffull) return late_fail(RT_EINVAL, "rt_render: strategy \"weighted\" with lights of mixed RNG use: the survey's records exceed 2^32 floats");
                rec_floats = size_t(*s->wt_total);
                fr.wt_recbase = s->wt_recbase;
            }
            rc = ensure(s, &s->wt_rec, &s->wt_rec_cap, rec_floats + 1); if (rc) return late_fail(rc, rt_last_error());
            rc = ensure(s, &s->wt_pick, &s->wt_pick_cap, size_t(n_points) + 1); if (rc) return late_fail(rc, rt_last_error());
            fr.wt_rec = s->wt_rec; fr.wt_pick = s->wt_pick;
            if (n_points > 0) {
                if ((rc = pass(2))) return rc;
                HIPCHK(hipEventRecord(s->wt_ev[2], s->stream));
                // LDS of the recurrence kernel: [state | chunk records | chunk picks] within 64 KB
                const size_t state_f = nL <= 64 ? 0 : size_t((3 * nL + 2) & ~1);
                int chunk = int((size_t(64) * 1024 / 4 - state_f - 2) / size_t(R + 2));
                chunk = chunk > 1024 ? 1024 : chunk < 1 ? 1 : chunk;
                const size_t lds = (state_f + size_t((chunk * R + 1) & ~1) + size_t(chunk) * 2) * sizeof(float);
                if (mixed) {
                    // LDS of the general form: [5 nL + 1 state | 2 x (group + 1) sample offsets | staged records | staged picks] within 64 KB
                    const int group = 256;
                    const size_t fixed = size_t(5 * nL + 2) + 2 * size_t(group + 1);
                    const size_t room = size_t(64) * 1024 / 4 - fixed - 4;
                    const int cap_points = int(room / 4 / 2), cap_floats = int(room - size_t(cap_points) * 2) & ~1;     // a quarter of the room for the picks (2 floats each)
                    const size_t mlds = (fixed + size_t(cap_floats) + 2 + size_t(cap_points) * 2) * sizeof(float);
                    hipLaunchKernelGGL(weighted_recurrence_mixed_kernel, dim3(1), dim3(64), mlds, s->stream, (const float *)s->wt_rec, s->wt_pick, (const unsigned *)s->wt_base,
                                       (const unsigned *)s->wt_recbase, (unsigned long long)fr.total_work, nL, nD, s->light_draw_flags, group, cap_floats, cap_points);
                } else {
                auto rk = nL <= 64 ? weighted_recurrence_lanes_kernel : weighted_recurrence_lds_kernel;      // light i in lane i | tables in LDS, one lane
                hipLaunchKernelGGL(rk, dim3(1), dim3(64), lds, s->stream, (const float *)s->wt_rec, s->wt_pick, n_points, nL, chunk);
                }
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(s->wt_ev[3], s->stream));
            } else { HIPCHK(hipEventRecord(s->wt_ev[2], s->stream)); HIPCHK(hipEventRecord(s->wt_ev[3], s->stream)); }
            if ((rc = pass(3))) return rc;
            HIPCHK(hipEventRecord(s->ev1, s->stream));
            s->last_pipeline = false; s->last_weighted = true;
        } else {
#ifdef RT_TAIL_PROBE
        static unsigned long long *probe = nullptr;
        const size_t n_waves = size_t(s->grids[variant]) * (RT_BLOCK / 64);
        if (!probe) HIPCHK(hipMalloc((void **)&probe, size_t(s->n_threads / 64 + 64) * 4 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(probe, 0, n_waves * 4 * sizeof(unsigned long long), s->stream));
        fr.probe = probe;
#endif
        HIPCHK(hipMemcpyAsync(s->dev_frame, &fr, sizeof(DevFrame), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemsetAsync(s->work_counter, 0, 64 * sizeof(unsigned long long), s->stream));
        HIPCHK(hipEventRecord(s->ev0, s->stream));
        hipLaunchKernelGGL(render_kernel_of(variant), dim3(s->grids[variant]), dim3(RT_BLOCK), 0, s->stream,
                           (const DevScene *)s->dev_scene, (const DevFrame *)s->dev_frame);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s->ev1, s->stream));
#ifdef RT_TAIL_PROBE
        {   // the launch's timeline: when the waves started, when each found the work list empty, when each ended (10 ns ticks of s_memrealtime)
            std::vector<unsigned long long> h(n_waves * 4);
            HIPCHK(hipStreamSynchronize(s->stream));
            HIPCHK(hipMemcpy(h.data(), probe, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t0max = 0, e0 = ~0ull, e1 = 0, x1 = 0, smin = ~0ull, smax = 0, stot = 0;
            std::vector<double> ends, drain;
            for (size_t w = 0; w < n_waves; ++w) {
                const unsigned long long *r = &h[4 * w];
                if (!r[2]) continue;
                t0 = std::min(t0, r[0]); t0max = std::max(t0max, r[0]); x1 = std::max(x1, r[2]);
                if (r[1]) { e0 = std::min(e0, r[1]); e1 = std::max(e1, r[1]); drain.push_back(double(r[2] - r[1]) * 1e-2); }
                smin = std::min(smin, r[3]); smax = std::max(smax, r[3]); stot += r[3];
                ends.push_back(double(r[2]));
            }
            std::sort(ends.begin(), ends.end()); std::sort(drain.begin(), drain.end());
            auto q = [&](const std::vector<double> &v, double f) { return v.empty() ? 0.0 : v[size_t(f * (v.size() - 1))]; };
            std::fprintf(stderr, "RT_TAIL_PROBE waves=%zu span=%.1f us | starts spread %.1f us | list first seen empty at %.1f us, last at %.1f us | wave ends (us after start): p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | "
                                 "drain per wave (end - empty, us): p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | samples per wave min %llu mean %.1f max %llu\n",
                         ends.size(), double(x1 - t0) * 1e-2, double(t0max - t0) * 1e-2, double(e0 - t0) * 1e-2, double(e1 - t0) * 1e-2,
                         (q(ends, .01) - double(t0)) * 1e-2, (q(ends, .1) - double(t0)) * 1e-2, (q(ends, .5) - double(t0)) * 1e-2, (q(ends, .9) - double(t0)) * 1e-2, (q(ends, .99) - double(t0)) * 1e-2, (q(ends, 1.) - double(t0)) * 1e-2,
                         q(drain, .1), q(drain, .5), q(drain, .9), q(drain, .99), q(drain, 1.), smin, double(stot) / double(std::max<size_t>(ends.size(), 1)), smax);
        }
#endif
        s->last_pipeline = false; s->last_weighted = false;
        }
    }
    if (!skip_film) { rc = film_gather_launch(s, fr, fg, (const DevFrame *)s->dev_frame, 0, fr.y_pixel_count); if (rc) return rc; }
    HIPCHK(hipEventRecord(s->ev2, s->stream));
    s->have_timing = true;
#ifdef RT_PROFILE
    {   // -DRT_PROFILE builds: per-wave cycle split of the render kernel
        unsigned long long v[24];
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(hipMemcpy(v, s->counters, sizeof v, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "RT_PROFILE shade_cyc=%llu trav_cyc=%llu outer=%llu inner=%llu rounds=%llu act_lane_rounds=%llu rays_at_trav_start=%llu desc_cyc=%llu leaf_cyc=%llu chunks=%llu pooled_rounds=%llu leaf_iters=%llu\n",
                     v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19]);
#ifdef RT_PROFILE_STAGES
        {   // cycles (wave-level, summed over waves) and lane visits per stage of the state machine (rt_integrate.h advance_pass)
            unsigned long long st[32];
            HIPCHK(hipMemcpy(st, s->counters + 24, sizeof st, hipMemcpyDeviceToHost));
            static const char *names[16] = {"FETCH", "VERTEX", "DIRECT_NEXT", "SHADOW_DONE", "MIS_DONE", "ED_BSDF", "ED_DONE", "BOUNCE", "SPECULAR", "SPEC_TRANS", "RETURN", "VOL_BEGIN", "VOL_STEP", "POP", "FINISH", "EXIT"};
            for (int i = 0; i < 16; ++i) if (st[2 * i + 1])
                std::fprintf(stderr, "RT_PROFILE_STAGE %-12s cycles=%llu passes=%llu lane_visits=%llu\n", names[i], st[2 * i], st[2 * i + 1] >> 40, st[2 * i + 1] & ((1ull << 40) - 1));
            HIPCHK(hipMemsetAsync(s->counters + 24, 0, 32 * sizeof(unsigned long long), s->stream));
        }
#endif
        HIPCHK(hipMemsetAsync(s->counters + 8, 0, 16 * sizeof(unsigned long long), s->stream));
    }
#endif
    return RT_OK;
}
int rt_sync(RtScene *s) {
    if (!s) return fail(RT_EINVAL, "null scene");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    return RT_OK;
}

int rt_counters(RtScene *s, RtCounters *out) {
    if (!s || !out) return fail(RT_EINVAL, "null argument");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    unsigned long long v[8];
    HIPCHK(hipMemcpy(v, s->counters, sizeof v, hipMemcpyDeviceToHost));
    out->camera_rays = v[0]; out->closest_rays = v[1]; out->any_rays = v[2]; out->nodes_visited = v[3];
    out->leaf_refs = v[4]; out->tri_tests = v[5]; out->bad_samples = v[6]; out->stack_overflows = v[7];
    return RT_OK;
}
int rt_counters_reset(RtScene *s) {
    if (!s) return fail(RT_EINVAL, "null scene");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemsetAsync(s->counters, 0, 24 * sizeof(unsigned long long), s->stream));
    return RT_OK;
}
int rt_set_counting(RtScene *s, int enabled) {
    if (!s) return fail(RT_EINVAL, "null scene");
    s->counting = enabled != 0;
    return RT_OK;
}
int rt_last_render_ms(RtScene *s, float *total_ms, float *kernel_ms) {
    if (!s || !s->have_timing) return fail(RT_ESTATE, "no timed launch yet");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipEventSynchronize(s->ev2));
    float k = 0.f, t = 0.f;
    HIPCHK(hipEventElapsedTime(&k, s->ev0, s->ev1));
    HIPCHK(hipEventElapsedTime(&t, s->ev0, s->ev2));
    if (total_ms) *total_ms = t;        // render kernel(s) + film gather
    if (kernel_ms) *kernel_ms = k;      // rt::render_kernel alone, or the whole shade / trace loop of the queue pipeline
    return RT_OK;
}
// Timing of the last rt_render by kernel: {whole frame, render part (megakernel or shade+trace loop), trace kernel launches summed,
// film gather} in ms, the number of pipeline iterations (0: megakernel) and how many of them were timed.
int rt_last_render_stats(RtScene *s, RtRenderStats *out) {
    if (!s || !out) return fail(RT_EINVAL, "null argument");
    if (!s->have_timing) return fail(RT_ESTATE, "no timed launch yet");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipEventSynchronize(s->ev2));
    std::memset(out, 0, sizeof *out);
    HIPCHK(hipEventElapsedTime(&out->render_ms, s->ev0, s->ev1));
    HIPCHK(hipEventElapsedTime(&out->total_ms, s->ev0, s->ev2));
    out->gather_ms = out->total_ms - out->render_ms;
    out->pipeline = s->last_pipeline ? 1 : 0;
    if (s->last_pipeline) {
        out->iterations = s->pipe_iters; out->timed_iterations = s->pipe_timed;
        float sum = 0.f;
        float sum2 = 0.f, sum3 = 0.f;
        for (int i = 0; i < s->pipe_timed; ++i) {
            float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, s->pipe_ev[6 * i], s->pipe_ev[6 * i + 1])); sum += ms;
            HIPCHK(hipEventElapsedTime(&ms, s->pipe_ev[6 * i + 2], s->pipe_ev[6 * i + 3])); sum2 += ms;
            if (s->last_marches) { HIPCHK(hipEventElapsedTime(&ms, s->pipe_ev[6 * i + 4], s->pipe_ev[6 * i + 5])); sum3 += ms; }
        }
        out->trace_ms = sum; out->shade_ms = sum2; out->march_ms = sum3;
        out->slots = s->pipe_slots;
    } else out->trace_ms = out->render_ms;
    out->bands = s->last_pipeline ? 0 : 1;
    if (s->last_weighted) {
        out->weighted_points = s->wt_points;
        hipEvent_t seq[6] = {s->ev0, s->wt_ev[0], s->wt_ev[1], s->wt_ev[2], s->wt_ev[3], s->ev1};
        for (int i = 0; i < 5; ++i) HIPCHK(hipEventElapsedTime(&out->weighted_ms[i], seq[i], seq[i + 1]));
    }
    return RT_OK;
}

}  // extern "C"
