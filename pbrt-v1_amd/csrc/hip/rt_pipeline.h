// rt_pipeline.h -- the queue ("wavefront") form of Scene::Render for scenes whose traversal is bound by memory latency.
//
// The megakernel (rt_render_kernel.h) keeps a lane's whole path state in registers while its ray is traced: ~128-160 VGPRs,
// 3-4 waves per SIMD, and at 1 M triangles only ~43 % of the lanes of a wave are traversing at any time.  Every one of the
// ~135 node visits of a ray is a dependent HBM / L2 round trip, so throughput is (rays in flight) / latency.  Here the same
// per-sample state machine (rt_integrate.h, unchanged arithmetic and draw order) is cut at its only blocking operation:
//
//   pipe_shade_kernel   one thread per path SLOT: load the slot's state (coalesced float4 planes), take the result of the ray
//                       it was waiting for, run advance_pass until the slot has set up its next ray (or fetched a new camera
//                       sample, or run out of work), append that ray to the closest-hit or the any-hit queue with one
//                       wave-aggregated atomic (ballot + popcount compaction: the queues hold live rays only, sorted by kind),
//                       store the state back.
//   pipe_trace_kernel   persistent waves that hold nothing but traversal state (<= 64 VGPRs -> 8 waves per SIMD): each lane
//                       pulls a ray from the queue, walks KdTreeAccel::Intersect / IntersectP (rt_traverse.h: same visit
//                       order, same counters), writes {prim, t, b1, b2} for its slot and immediately pulls the next ray, so
//                       all 64 lanes traverse all the time and twice as many waves hide the latency.
//
// The host alternates the two kernels until a shade pass enqueues nothing.  Cost: ~400 B of state traffic per ray (coalesced,
// 100 M rays -> 40 GB = ~7 ms at HBM rate) against ~1.5 KB of scattered traversal traffic per ray.  Whitted / DirectLighting
// recursion frames and the volume scratch are indexed by slot instead of by resident thread.
#pragma once
#include "rt_integrate.h"

#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 6          // waves per SIMD the trace kernel is built for: its 24 KB of LDS stack planes allow 6 workgroups per CU (80 VGPRs)
#endif
#ifndef RT_TRACE_STACK
#define RT_TRACE_STACK 8          // LDS ring entries per lane in the trace kernel: 8 x 8 B x 256 = 16 KB per workgroup, 8 workgroups per CU
#endif
#ifndef RT_TRACE_REFILL
#define RT_TRACE_REFILL 16        // a wave refills its idle lanes from the queue when at least this many are idle
#endif
#ifndef RT_TRACE_CHUNK_MAX
#define RT_TRACE_CHUNK_MAX 256    // queue rays a trace wave takes from the global head at a time: remaining / (2 * waves), clamped
#endif
#ifndef RT_TRACE_CHUNK_MIN
#define RT_TRACE_CHUNK_MIN 32
#endif
#ifndef RT_PIPE_TRACE_DSTEPS
#define RT_PIPE_TRACE_DSTEPS 1    // steps per round in the trace kernel (the megakernel takes RT_TRACE_DSTEPS = 2): C5's trace launches 203 -> 195 ms
#endif

namespace rt {

static_assert(ST_EXIT < 16, "the slot's control word holds the stage in 4 bits");
#define RT_PIPE_VEC 12            // float4 planes of slot state (the 11th only for DirectLighting "all", the 12th for the EXT kernels)

struct PipePool {
    unsigned n_slots;
    float4 *state;                // [RT_PIPE_VEC][n_slots]
    float4 *ray_o, *ray_d;        // [n_slots] {o, mint} {d, maxt}: the slot's ray, kept for the shading that follows the trace
    float4 *hit;                  // [n_slots] {prim, t, b1, b2} written by the trace kernel
    float4 *q_o, *q_d;            // [2][n_slots] compacted queues: closest-hit rays in [0, n), any-hit rays in [n_slots, n_slots + m)
    unsigned *q_slot;             // [2][n_slots] slot of each queued ray
    unsigned *q_march;            // [n_slots] frames with a medium: the slots parked in ST_VOL_STEP, whose ray march rt::pipe_march_kernel runs
    unsigned *q_count;            // [iterations][RT_QC_STRIDE]: n closest at +0, consumer head at +RT_QC_HEAD, n any at +RT_QC_ANY
                                  // (three different 64-byte lines: each is the target of one kernel's atomics)
    unsigned long long *wave_work; // [n_slots / 64][2] {next, end}: the chunk of camera samples a wave of the shade kernel owns
};
// one launch of a shade kernel: which part of the pool it covers and where its rays go.  The pool can be run as two halves that take turns
// (one half is shaded on its own set of CUs while the other's rays are traced, rt_kernels.hip render_pipeline)
struct PipeLaunch {
    unsigned qi;                  // index of this launch's counters in q_count
    unsigned slot_base;           // first slot of the launch
    unsigned q_base;              // first entry of its region of the queue arrays
};
#define RT_QC_STRIDE 64
#define RT_QC_HEAD 16
#define RT_QC_ANY 32
#define RT_QC_MARCH 48            // number of parked marches of the iteration, the march kernel's consumer head behind it
#define RT_QC_MHEAD 56
#define RT_WORK_CHUNK 128         // camera samples a shade wave takes from the global work counter at a time

// ---- slot state <-> Lane ----------------------------------------------------------------------------------------
// ctl word: stage (4 bits) | has_ray << 4 | specular << 5 | any << 6 | depth << 8 | fsp << 16
template <int INTEG, bool EXT>
RT_DEV void pipe_load(const PipePool &pl, const DevFrame &fr, unsigned slot, Lane &ln) {
    const float4 RT_G *st = RT_GPTR(const float4, pl.state) + slot;
    const size_t n = pl.n_slots;
    const float4 a0 = st[0], a1 = st[n];
    const unsigned ctl = __float_as_uint(a1.z);
    ln.stage = int(ctl & 15u); ln.has_ray = (ctl >> 4) & 1u; ln.specular = (ctl >> 5) & 1u; ln.tv.any = (ctl >> 6) & 1u;
    ln.depth = int((ctl >> 8) & 255u); ln.fsp = int(ctl >> 16);
    ln.sample_index = __float_as_uint(a0.x); ln.work = __float_as_uint(a0.y); ln.image_x = a0.z; ln.image_y = a0.w;
    ln.dim_base = __float_as_uint(a1.x); ln.rng.ctr = __float_as_uint(a1.y); ln.alpha = a1.w;
    ln.rng.base = rng_base(ln.sample_index, fr.seed);
    ln.tv.active = false; ln.tv.hit_prim = -1;
    if (ln.stage == ST_EXIT || ln.stage == ST_FETCH) {        // nothing else is live
        ln.L = mk3(0.f); ln.thr = mk3(1.f);
        return;
    }
    if (ln.stage == ST_POP) {                                  // back from rt::pipe_march_kernel: Scene::Li of this level is complete in L; everything else the
        const float4 a2 = st[2 * n];                           // slot needs from here on is in the control planes (fsp == 0) or in its recursion frame (frame_pop)
        ln.L = mk3(a2.x, a2.y, a2.z);
        { const unsigned ml = __float_as_uint(a2.w); ln.v.mat = int(ml & 0xffffu); ln.v.light = int(ml >> 16) - 1; }
        ln.thr = mk3(1.f);
        return;
    }
    if (ln.stage == ST_VERTEX) {                               // waiting for a camera / continuation / specular ray: the next vertex is made from the hit (make_vertex) and
        const float4 a2 = st[2 * n], a3 = st[3 * n];           // every field of the direct-lighting loop is set before it is read (stage_body ST_VERTEX, ST_DIRECT_NEXT): only
        ln.L = mk3(a2.x, a2.y, a2.z);                          // the radiance bookkeeping travels (round 6: 4 planes instead of 11 each way, a quarter of a C5 sample's state traffic)
        { const unsigned ml = __float_as_uint(a2.w); ln.v.mat = int(ml & 0xffffu); ln.v.light = int(ml >> 16) - 1; }
        ln.thr = mk3(a3.x, a3.y, a3.z);
        { const unsigned lj = __float_as_uint(a3.w); ln.li = int(lj & 0xffffu); ln.lj = int(lj >> 16); }
        return;
    }
    const float4 a2 = st[2 * n], a3 = st[3 * n], a4 = st[4 * n], a5 = st[5 * n], a6 = st[6 * n], a7 = st[7 * n], a8 = st[8 * n], a9 = st[9 * n];
    ln.L = mk3(a2.x, a2.y, a2.z);
    { const unsigned ml = __float_as_uint(a2.w); ln.v.mat = int(ml & 0xffffu); ln.v.light = int(ml >> 16) - 1; }
    ln.thr = mk3(a3.x, a3.y, a3.z);
    { const unsigned lj = __float_as_uint(a3.w); ln.li = int(lj & 0xffffu); ln.lj = int(lj >> 16); }
    ln.v.p = mk3(a4.x, a4.y, a4.z); ln.cur_light = __float_as_int(a4.w);
    ln.v.nn = mk3(a5.x, a5.y, a5.z); ln.bs1 = a5.w;
    ln.v.sn = mk3(a6.x, a6.y, a6.z); ln.bs2 = a6.w;
    ln.v.tn = cross3(ln.v.nn, ln.v.sn);                        // make_vertex's own expression (rt_shade.h)
    ln.v.wo = mk3(a7.x, a7.y, a7.z); ln.bcs = a7.w;
    ln.Ld = mk3(a8.x, a8.y, a8.z);
    ln.pend = mk3(a8.w, a9.x, a9.y);
    if (INTEG == RT_INTEGRATOR_DIRECT) {
        const float4 a10 = st[10 * n];
        ln.Ld_light = mk3(a9.z, a9.w, a10.x); ln.L_all = mk3(a10.y, a10.z, a10.w);
    }
    if (EXT) { const float4 a11 = st[11 * n]; ln.v.ng = mk3(a11.x, a11.y, a11.z); }
}
template <int INTEG, bool EXT>
RT_DEV void pipe_store(const PipePool &pl, unsigned slot, const Lane &ln) {
    float4 RT_G *st = RT_GPTR(float4, pl.state) + slot;
    const size_t n = pl.n_slots;
    const unsigned ctl = unsigned(ln.stage) | (ln.has_ray ? 16u : 0u) | (ln.specular ? 32u : 0u) | (ln.tv.any ? 64u : 0u) |
                         (unsigned(ln.depth) << 8) | (unsigned(ln.fsp) << 16);
    st[0] = make_float4(__uint_as_float(ln.sample_index), __uint_as_float(ln.work), ln.image_x, ln.image_y);
    st[n] = make_float4(__uint_as_float(ln.dim_base), __uint_as_float(ln.rng.ctr), __uint_as_float(ctl), ln.alpha);
    if (ln.stage == ST_EXIT) return;
    st[2 * n] = make_float4(ln.L.x, ln.L.y, ln.L.z, __uint_as_float(unsigned(ln.v.mat) | (unsigned(ln.v.light + 1) << 16)));
    if (ln.stage == ST_VOL_STEP) return;                       // parked for the march kernel: the surface vertex is dead (see pipe_load)
    st[3 * n] = make_float4(ln.thr.x, ln.thr.y, ln.thr.z, __uint_as_float(unsigned(ln.li) | (unsigned(ln.lj) << 16)));
    if (ln.stage == ST_VERTEX) return;                         // (only ever stored while the slot waits for the ray that leads to its next vertex: see pipe_load)
    st[4 * n] = make_float4(ln.v.p.x, ln.v.p.y, ln.v.p.z, __int_as_float(ln.cur_light));
    st[5 * n] = make_float4(ln.v.nn.x, ln.v.nn.y, ln.v.nn.z, ln.bs1);
    st[6 * n] = make_float4(ln.v.sn.x, ln.v.sn.y, ln.v.sn.z, ln.bs2);
    st[7 * n] = make_float4(ln.v.wo.x, ln.v.wo.y, ln.v.wo.z, ln.bcs);
    st[8 * n] = make_float4(ln.Ld.x, ln.Ld.y, ln.Ld.z, ln.pend.x);
    if (INTEG == RT_INTEGRATOR_DIRECT) {
        st[9 * n] = make_float4(ln.pend.y, ln.pend.z, ln.Ld_light.x, ln.Ld_light.y);
        st[10 * n] = make_float4(ln.Ld_light.z, ln.L_all.x, ln.L_all.y, ln.L_all.z);
    } else st[9 * n] = make_float4(ln.pend.y, ln.pend.z, 0.f, 0.f);
    if (EXT) st[11 * n] = make_float4(ln.v.ng.x, ln.v.ng.y, ln.v.ng.z, 0.f);
}

// ---- shade: everything between two rays of a path, for every slot ---------------------------------------------------
template <bool COUNT, int INTEG, bool VOL, bool EXT>
__global__ __launch_bounds__(RT_BLOCK, 1) void pipe_shade_kernel(const DevScene *__restrict__ scp, const DevFrame *__restrict__ frp,
                                                               const PipePool *__restrict__ plp, PipeLaunch pk) {
    const DevScene &sc = *scp;
    const DevFrame &fr = *frp;
    const PipePool &pl = *plp;
    const unsigned slot = pk.slot_base + blockIdx.x * RT_BLOCK + threadIdx.x;   // slot_base and n_slots are multiples of RT_BLOCK
    const int lane = threadIdx.x & 63;
    Lane ln;
    ln.v.p = ln.v.nn = ln.v.ng = ln.v.sn = ln.v.tn = ln.v.wo = mk3(0.f); ln.v.mat = 0; ln.v.light = -1;
    ln.li = ln.lj = 0; ln.cur_light = 0; ln.Ld = ln.Ld_light = ln.L_all = ln.pend = mk3(0.f); ln.bs1 = ln.bs2 = ln.bcs = 0.f;
    pipe_load<INTEG, EXT>(pl, fr, slot, ln);
    if (!__syncthreads_or(ln.stage != ST_EXIT)) return;
    __shared__ unsigned blk_cnt[3], blk_base[3];
    if (threadIdx.x < 3) blk_cnt[threadIdx.x] = 0u;
    // this wave's chunk of the work list (wave-uniform): 64 K waves hammering ONE counter cost more than the shading itself
    // (measured: 1.1 ms per pass, ~0.15 ms of it arithmetic); a wave now goes to the global counter once per RT_WORK_CHUNK samples
    unsigned long long RT_G *ww = RT_GPTR(unsigned long long, pl.wave_work) + size_t(slot >> 6) * 2;
    unsigned long long w_next = ww[0], w_end = ww[1];
    if (ln.has_ray) {                                                   // the ray this slot was waiting for has been traced
        const float4 ro = RT_GPTR(const float4, pl.ray_o)[slot], rd = RT_GPTR(const float4, pl.ray_d)[slot];
        const float4 h = RT_GPTR(const float4, pl.hit)[slot];
        ln.tv.o = mk3(ro.x, ro.y, ro.z); ln.tv.mint = ro.w; ln.tv.d = mk3(rd.x, rd.y, rd.z); ln.tv.maxt = rd.w;
        ln.tv.hit_prim = __float_as_int(h.x);
        if (ln.tv.hit_prim >= 0 && !ln.tv.any) ln.tv.maxt = h.y;        // primitive.cpp:120: the accepted hit shortens the ray
        ln.tv.b1 = h.z; ln.tv.b2 = h.w;
        ln.has_ray = false;
    }
    unsigned c_cam = 0, c_closest = 0, c_any = 0, c_bad = 0;
    do {
        advance_pass<COUNT, INTEG, VOL, EXT, true, true>(sc, fr, ln, slot, &c_closest, &c_any, &c_bad, -1);
        const unsigned long long want = __ballot(!ln.has_ray && ln.stage == ST_FETCH);
        if (want) {                                                     // work fetch from the wave's chunk
            const unsigned n_want = unsigned(__popcll(want));
            const unsigned long long have = w_end - w_next;             // what is left of the chunk is used first, the rest comes from a fresh one
            unsigned long long fresh = 0;
            if (have < n_want) {                                        // wave-uniform branch
                if (lane == 0) fresh = atomicAdd(fr.work_counter, (unsigned long long)RT_WORK_CHUNK);
                fresh = __shfl(fresh, 0);
            }
            if (!ln.has_ray && ln.stage == ST_FETCH) {
                const unsigned long long r = __popcll(want & ((1ull << lane) - 1ull));
                const unsigned long long w = r < have ? w_next + r : fresh + (r - have);
                if (w >= fr.total_work) ln.stage = ST_EXIT;
                else {
                    unsigned long long pixel; int s;
                    if (work_to_sample(fr, w, pixel, s)) {
                        Ray ray;
                        setup_sample(sc, fr, ln, pixel, s, ray);
                        ln.work = uint32_t(w);
                        ln.L = mk3(0.f); ln.thr = mk3(1.f); ln.alpha = 0.f; ln.depth = 0; ln.fsp = 0;
                        ln.specular = false;
                        if (COUNT) ++c_cam;
                        if (VOL) vol_store_ray(fr, 0, slot, ray);
                        launch_ray<true>(ln, sc, ray.o, ray.d, ray.mint, ray.maxt, false, ST_VERTEX);
                    }
                }
            }
            if (have < n_want) { w_next = fresh + (n_want - have); w_end = fresh + RT_WORK_CHUNK; }
            else w_next += n_want;
        }
    } while (__any(!ln.has_ray && ln.stage != ST_EXIT && !(VOL && ln.stage == ST_VOL_STEP)));      // (a slot in ST_VOL_STEP without a ray is parked for the march kernel)
    if (lane == 0) { ww[0] = w_next; ww[1] = w_end; }
    // ---- enqueue: ballot compaction per ray kind inside the wave, LDS atomics inside the workgroup, ONE global atomic per
    // workgroup and kind (the two counters sit on different cache lines)
    {
        unsigned RT_G *qc = RT_GPTR(unsigned, pl.q_count) + size_t(pk.qi) * RT_QC_STRIDE;
        const unsigned long long mc = __ballot(ln.has_ray && !ln.tv.any), ma = __ballot(ln.has_ray && ln.tv.any);
        const bool parked = VOL && !ln.has_ray && ln.stage == ST_VOL_STEP;
        const unsigned long long mm = __ballot(parked);
        unsigned wc = 0, wa = 0, wm = 0;
        __syncthreads();                                                // blk_cnt zeroed
        if (lane == 0) {
            if (mc) wc = atomicAdd(&blk_cnt[0], unsigned(__popcll(mc)));
            if (ma) wa = atomicAdd(&blk_cnt[1], unsigned(__popcll(ma)));
            if (VOL && mm) wm = atomicAdd(&blk_cnt[2], unsigned(__popcll(mm)));
        }
        __syncthreads();
        if (threadIdx.x == 0 && blk_cnt[0]) blk_base[0] = atomicAdd((unsigned *)qc, blk_cnt[0]);
        if (threadIdx.x == 64 && blk_cnt[1]) blk_base[1] = atomicAdd((unsigned *)(qc + RT_QC_ANY), blk_cnt[1]);
        if (VOL && threadIdx.x == 128 && blk_cnt[2]) blk_base[2] = atomicAdd((unsigned *)(qc + RT_QC_MARCH), blk_cnt[2]);
        __syncthreads();
        wc = __shfl(wc, 0) + blk_base[0]; wa = __shfl(wa, 0) + blk_base[1];
        const unsigned wm0 = VOL ? __shfl(wm, 0) + blk_base[2] : 0u;
        if (VOL && parked) RT_GPTR(unsigned, pl.q_march)[size_t(pk.q_base) + wm0 + __popcll(mm & ((1ull << lane) - 1ull))] = slot;
        if (ln.has_ray) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const size_t q = size_t(pk.q_base) + (ln.tv.any ? size_t(pl.n_slots) + wa + __popcll(ma & below) : size_t(wc) + __popcll(mc & below));
            const float4 ro = make_float4(ln.tv.o.x, ln.tv.o.y, ln.tv.o.z, ln.tv.mint), rd = make_float4(ln.tv.d.x, ln.tv.d.y, ln.tv.d.z, ln.tv.maxt);
            RT_GPTR(float4, pl.q_o)[q] = ro; RT_GPTR(float4, pl.q_d)[q] = rd; RT_GPTR(unsigned, pl.q_slot)[q] = slot;
            RT_GPTR(float4, pl.ray_o)[slot] = ro; RT_GPTR(float4, pl.ray_d)[slot] = rd;
        }
    }
    pipe_store<INTEG, EXT>(pl, slot, ln);
    if (COUNT) {
        unsigned long long v[4] = {c_cam, c_closest, c_any, c_bad};
        const int idx[4] = {0, 1, 2, 6};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0 && x) atomicAdd(fr.counters + idx[k], x);
        }
    }
}

// ---- trace: persistent waves, ray replacement -------------------------------------------------------------------------
struct TraceJob {
    const float4 *q_o, *q_d;       // queue rays: closest in [0, n_slots), any in [n_slots, 2 n_slots)
    const unsigned *q_slot;
    int by_slot;                   // rt_pipe_vertex.h: q_slot holds slot | kind << 30; q_o = ray origins [slot], q_d = directions [kind][slot],
                                   // hit = [kind][slot]; kind 0 is an any-hit ray
    unsigned q_base;               // where this launch's part of the queue arrays starts (the pool's halves have their own queue regions)
    unsigned *q_count;             // this iteration's counters: n closest at +0, head at +RT_QC_HEAD, n any at +RT_QC_ANY
    float4 *hit;                   // [slot]; for rt_trace_* (q_slot == nullptr) indexed by queue position
    unsigned n_slots;
    uint2 *spill; unsigned n_threads;
    unsigned long long *counters;
};

template <bool COUNT, int ACCEL, bool EXT>
__global__ __launch_bounds__(RT_BLOCK, RT_TRACE_WAVES) void pipe_trace_kernel(const DevScene *__restrict__ scp, TraceJob job) {
    __shared__ uint2 lds_stack[RT_TRACE_STACK * RT_BLOCK];
    __shared__ float lds_tm[(ACCEL != RT_ACCEL_GRID && !EXT) ? RT_TRACE_STACK * RT_BLOCK : 1];       // second plane of the pair-form stack
    const DevScene &sc = *scp;
    const unsigned gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned n_closest = job.q_count[0], total = n_closest + job.q_count[RT_QC_ANY];
    TravCounters tc; tc.nodes = tc.leaf_refs = tc.tris = tc.spills = 0;
    Trav tv; tv.active = false; tv.at_leaf = false; tv.hit_prim = -1; tv.any = false; tv.maxt = 0.f; tv.b1 = tv.b2 = 0.f;
    unsigned slot = 0; bool busy = false;
    bool exhausted = false;
    unsigned w_next = 0, w_end = 0;                                     // this wave's chunk of the queue (wave-uniform)
    bool head_done = false;                                             // the global head has passed the end of the queue
    unsigned w_seen = 0;
    const unsigned n_waves = gridDim.x * (RT_BLOCK / 64);
    // Outer loop: report finished rays, refill the idle lanes from the queue (one wave-aggregated atomic).  Inner loop: rounds of
    // traversal with nothing else in it, until enough lanes have finished for a refill to pay (or, once the queue is exhausted,
    // until the wave's last ray ends).
#pragma unroll 1
    for (;;) {
        if (busy && !tv.active) {
            RT_GPTR(float4, job.hit)[slot] = make_float4(__int_as_float(tv.hit_prim), tv.hit_prim >= 0 ? tv.maxt : 0.f, tv.b1, tv.b2);
            busy = false;
        }
        const unsigned long long idle = __ballot(!busy);
        const unsigned n_idle = unsigned(__builtin_amdgcn_readfirstlane(__popcll(idle)));
        if (!exhausted && n_idle >= RT_TRACE_REFILL) {
            // Rays come from the wave's private chunk of the queue; the global head is touched once per chunk (one device-scope counter
            // serves ~6-8 ns per atomic whoever asks: with one atomic per refill the head, not the traversal, set this kernel's pace below
            // 32 idle lanes per refill -- profiles/r02_scan_util3.jsonl).  Chunks shrink as the queue drains (guided self-scheduling), so
            // the launch's tail stays a few rays per wave.
            const unsigned have = w_end - w_next;                       // w_end never exceeds total
            unsigned f_lo = 0, f_hi = 0;                                 // the fresh chunk, clipped to the queue
            if (have < n_idle && !head_done) {                          // wave-uniform branch
                const int leader = __ffsll((long long)idle) - 1;
                unsigned want = (total > w_seen ? total - w_seen : 0u) / (2u * n_waves + 1u);     // w_seen: the head as this wave last saw it
                want = want < RT_TRACE_CHUNK_MIN ? RT_TRACE_CHUNK_MIN : (want > RT_TRACE_CHUNK_MAX ? RT_TRACE_CHUNK_MAX : want);
                want = want < n_idle - have ? n_idle - have : want;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(job.q_count + RT_QC_HEAD, want);
                base = __builtin_amdgcn_readfirstlane(__shfl(base, leader));
                if (base < total) { f_lo = base; f_hi = base + want < total ? base + want : total; w_seen = base + want; }
                else head_done = true;
            }
            const unsigned rk = unsigned(__popcll(idle & ((1ull << lane) - 1ull)));
            const unsigned fi = f_lo + (rk - have);
            const unsigned i = rk < have ? w_next + rk : (fi < f_hi ? fi : total);     // total = no ray for this lane
            if (have < n_idle) {
                const unsigned took = n_idle - have < f_hi - f_lo ? n_idle - have : f_hi - f_lo;
                if (f_hi > f_lo) { w_next = f_lo + took; w_end = f_hi; } else w_next = w_end;
            } else w_next += n_idle;
            w_next = __builtin_amdgcn_readfirstlane(w_next); w_end = __builtin_amdgcn_readfirstlane(w_end);      // wave-uniform: scalar registers
            exhausted = head_done && w_next >= w_end;
            if (!busy && i < total) {
                bool any = i >= n_closest;
                size_t q = any ? size_t(job.n_slots) + job.q_base + (i - n_closest) : size_t(job.q_base) + i;
                size_t qo = q;
                if (job.by_slot) {                                      // wave-uniform
                    const unsigned e = RT_GPTR(const unsigned, job.q_slot)[size_t(job.q_base) + i];
                    qo = e & 0x3fffffffu; q = size_t(e >> 30) * job.n_slots + qo; any = (e >> 30) == 0u;
                }
                const float4 ro = RT_GPTR(const float4, job.q_o)[qo], rd = RT_GPTR(const float4, job.q_d)[q];
                slot = job.by_slot ? unsigned(q) : (job.q_slot ? RT_GPTR(const unsigned, job.q_slot)[q] : unsigned(q));
                Ray r; r.o = mk3(ro.x, ro.y, ro.z); r.mint = ro.w; r.d = mk3(rd.x, rd.y, rd.z); r.maxt = rd.w;
                accel_begin<ACCEL>(tv, sc, r, any);
                busy = true;
            }
        }
        const int live0 = __popcll(__ballot(busy && tv.active));
        if (live0 == 0) { if (exhausted && !__any(busy)) break; else if (exhausted) continue; }
        // leave the inner loop when RT_TRACE_REFILL more lanes have finished (queue not exhausted) or when all have
        const int leave_at = exhausted ? 0 : (live0 > RT_TRACE_REFILL ? live0 - RT_TRACE_REFILL : 0);
#pragma unroll 1
        do {
            trace_round<COUNT, ACCEL, EXT, RT_TRACE_STACK, true, RT_PIPE_TRACE_DSTEPS>(tv, busy, sc, (uint2 RT_L *)lds_stack, (float RT_L *)lds_tm, RT_GPTR(uint2, job.spill), job.n_threads, gtid, tc);
        } while (__popcll(__ballot(busy && tv.active)) > leave_at);
    }
    if (COUNT) {
        unsigned long long v[4] = {tc.nodes, tc.leaf_refs, tc.tris, tc.spills};
        const int idx[4] = {3, 4, 5, 7};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0 && x) atomicAdd(job.counters + idx[k], x);
        }
    }
}

typedef void (*PipeShadeFn)(const DevScene *, const DevFrame *, const PipePool *, PipeLaunch);
typedef void (*PipeTraceFn)(const DevScene *, TraceJob);

}  // namespace rt
