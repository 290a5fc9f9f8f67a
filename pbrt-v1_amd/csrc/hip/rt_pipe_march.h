// rt_pipe_march.h -- the ray-march kernel of the queue pipeline (frames with a participating medium).
//
// Scene::Li = T * Lo + Lv (scene.cpp:120-126): after the surface integrator has finished a level, the volume integrator marches along that
// level's ray -- EmissionIntegrator::Li emission.cpp:60-95, SingleScattering::Li single.cpp:57-116 -- and single scattering casts ONE shadow
// ray per step.  Round 3 ran a march as one pipeline iteration per step: every step loaded and stored the slot's state (5 + 5 float4 planes,
// 13 + 13 floats of march state, ray and hit records: ~450 B per shadow ray, 400-500 GB per C5 frame, 13-17x what SURVEY 8(d) prices shading
// at) and the frame took as many shade / trace launch pairs as its longest march has steps (65).
// Here a march's steps run to completion inside ONE persistent kernel.  The shade kernel runs the head of the march (march_begin: clip, step size,
// the LatinHypercube table -- 6 N dependent memory operations, hidden only where millions of slots do them side by side), parks the slot
// (advance_pass PARK) and queues its id; a lane of this kernel takes a parked slot, keeps the march -- {i, N, t0, step, Tr, p, Lv}, the level's ray, the sample's
// draw counter -- in registers, and between steps traces the step's shadow ray in the wave's shared traversal loop (trace_round: the very
// code of pipe_trace_kernel, KdTreeAccel::IntersectP kdtree.cpp:404-488), exactly as the megakernel does for whole paths; a lane whose march
// ends writes L = T * L + Lv and the counter back (two float4s), marks the slot ST_POP and takes the next parked slot.  State traffic per march:
// 132 B in, 32 B out (+ 12 B of its sample table per step), whatever its length.  The arithmetic and the order of RandomFloat() draws are march_begin's and march_steps' (rt_integrate.h), shared with the
// megakernel: films are bit-identical (tests/test_gpu_configs.py).
#pragma once
#include "rt_pipeline.h"

#ifndef RT_MARCH_WAVES
#define RT_MARCH_WAVES 4          // waves per SIMD the march kernel is built for (128 VGPRs)
#endif
#ifndef RT_MARCH_STACK
#define RT_MARCH_STACK 8          // LDS ring entries per lane (pair form: 12 B each): 24 KB per workgroup
#endif
#ifndef RT_MARCH_REFILL
#define RT_MARCH_REFILL 16        // leave the traversal loop when this many more lanes hold a finished shadow ray (they take their next step): 8 and 32 measured slower
#endif
#ifndef RT_MARCH_TAKE
#define RT_MARCH_TAKE 8           // idle lanes (march complete) take new slots from the queue when at least this many are idle (C5: 16: 216 ms of march launches, 8: 208, 4: 208, 32: 247)
#endif

namespace rt {

static_assert(RT_MARCH_STACK >= RT_TRACE_STACK, "the shared spill area is sized for RT_TRACE_STACK ring entries");

struct MarchJob {
    const unsigned *q_march;       // the slots parked by this iteration's shade pass
    unsigned *q_count;             // this iteration's counters: marches at +RT_QC_MARCH, this kernel's consumer head at +RT_QC_MHEAD
    uint2 *spill; unsigned n_threads;       // traversal-stack spill area [entry][thread]
    unsigned long long *counters;
};

template <bool COUNT, int ACCEL, bool EXT>
__global__ __launch_bounds__(RT_BLOCK, COUNT ? 1 : RT_MARCH_WAVES) void pipe_march_kernel(const DevScene *__restrict__ scp, const DevFrame *__restrict__ frp,
                                                                             const PipePool *__restrict__ plp, MarchJob job) {
    __shared__ uint2 lds_stack[RT_MARCH_STACK * RT_BLOCK];
    __shared__ float lds_tm[(ACCEL != RT_ACCEL_GRID && !EXT) ? RT_MARCH_STACK * RT_BLOCK : 1];
    const DevScene &sc = *scp;
    const DevFrame &fr = *frp;
    const PipePool &pl = *plp;
    const unsigned gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned total = job.q_count[RT_QC_MARCH];
    const size_t n = pl.n_slots;
    TravCounters tc; tc.nodes = tc.leaf_refs = tc.tris = tc.spills = 0;
    unsigned c_any = 0;
    Lane ln;
    ln.has_ray = false; ln.tv.active = false; ln.tv.at_leaf = false; ln.tv.hit_prim = -1; ln.tv.any = true; ln.tv.maxt = 0.f; ln.tv.b1 = ln.tv.b2 = 0.f;
    ln.L = mk3(0.f); ln.pend = mk3(0.f); ln.stage = ST_EXIT; ln.fsp = 0;
    March m; m.i = m.N = 0; m.t0 = m.step = 0.f; m.Tr = m.p = m.Lv = mk3(0.f); m.s0 = m.s1 = m.s2 = 0.f;
    Ray ray; ray.o = ray.d = mk3(0.f); ray.mint = ray.maxt = 0.f;
    unsigned slot = 0, ctl = 0;
    bool busy = false;                                                  // this lane holds a march
    bool exhausted = false, head_done = false;
    unsigned w_next = 0, w_end = 0, w_seen = 0;                         // this wave's chunk of the march queue (wave-uniform)
    const unsigned n_waves = gridDim.x * (RT_BLOCK / 64);
#pragma unroll 1
    for (;;) {
        // ---- idle lanes take parked slots from the wave's chunk of the queue (guided chunks, one atomic per chunk: as pipe_trace_kernel)
        bool begin = false;
        const unsigned long long idle = __ballot(!busy);
        const unsigned n_idle = unsigned(__builtin_amdgcn_readfirstlane(__popcll(idle)));
        if (!exhausted && n_idle >= RT_MARCH_TAKE) {
            const unsigned have = w_end - w_next;
            unsigned f_lo = 0, f_hi = 0;
            if (have < n_idle && !head_done) {                          // wave-uniform branch
                const int leader = __ffsll((long long)idle) - 1;
                unsigned want = (total > w_seen ? total - w_seen : 0u) / (2u * n_waves + 1u);
                want = want < 64u ? 64u : (want > 256u ? 256u : want);
                want = want < n_idle - have ? n_idle - have : want;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(job.q_count + RT_QC_MHEAD, want);
                base = __builtin_amdgcn_readfirstlane(__shfl(base, leader));
                if (base < total) { f_lo = base; f_hi = base + want < total ? base + want : total; w_seen = base + want; }
                else head_done = true;
            }
            const unsigned rk = unsigned(__popcll(idle & ((1ull << lane) - 1ull)));
            const unsigned fi = f_lo + (rk - have);
            const unsigned i = rk < have ? w_next + rk : (fi < f_hi ? fi : total);
            if (have < n_idle) {
                const unsigned took = n_idle - have < f_hi - f_lo ? n_idle - have : f_hi - f_lo;
                if (f_hi > f_lo) { w_next = f_lo + took; w_end = f_hi; } else w_next = w_end;
            } else w_next += n_idle;
            w_next = __builtin_amdgcn_readfirstlane(w_next); w_end = __builtin_amdgcn_readfirstlane(w_end);
            exhausted = head_done && w_next >= w_end;
            if (!busy && i < total) {
                slot = RT_GPTR(const unsigned, job.q_march)[i];
                const float4 RT_G *st = RT_GPTR(const float4, pl.state) + slot;
                const float4 a0 = st[0], a1 = st[n], a2 = st[2 * n];
                ctl = __float_as_uint(a1.z);
                ln.fsp = int(ctl >> 16);
                ln.rng.ctr = __float_as_uint(a1.y);                        // (the sampler dimensions were consumed by march_begin: only the draw counter goes on)
                ln.rng.base = rng_base(__float_as_uint(a0.x), fr.seed);
                ln.L = mk3(a2.x, a2.y, a2.z);
                ray = vol_load_ray(fr, ln.fsp, slot);
                const float RT_G *vs = RT_GPTR(const float, fr.vol_state) + slot;         // what march_begin left (rt_integrate.h stage_body, PARK)
                m.i = __float_as_int(vs[0]); m.N = __float_as_int(vs[n]); m.t0 = vs[2 * n]; m.step = vs[3 * n];
                m.Tr = mk3(vs[4 * n], vs[5 * n], vs[6 * n]); m.p = mk3(vs[7 * n], vs[8 * n], vs[9 * n]); m.Lv = mk3(vs[10 * n], vs[11 * n], vs[12 * n]);
                ln.has_ray = false;
                busy = true; begin = true;
            }
        }
        // ---- marching: a lane with a new slot starts its march, a lane whose shadow ray has ended takes the result and marches on; both
        // come back either with the next step's shadow ray set up or with the level complete
        const bool resume = busy && !begin && ln.has_ray && !ln.tv.active;
        if (begin || resume) {
            ln.has_ray = false;
            if (!march_steps<COUNT, EXT, false, true>(sc, fr, ln, ray, m, RT_GPTR(const float, fr.vol_samp) + slot, n, resume, &c_any)) {
                float RT_G *c1 = (float RT_G *)(RT_GPTR(float4, pl.state) + slot + n);
                float RT_G *c2 = (float RT_G *)(RT_GPTR(float4, pl.state) + slot + 2 * n);
                c1[1] = __uint_as_float(ln.rng.ctr); c1[2] = __uint_as_float((ctl & ~0x1fu) | unsigned(ST_POP));     // the draw counter, the control word (now ST_POP)
                c2[0] = ln.L.x; c2[1] = ln.L.y; c2[2] = ln.L.z;
                busy = false;
            }
        }
        // ---- the shared traversal loop (IntersectP of the steps' shadow rays)
        const int live0 = __popcll(__ballot(busy && ln.has_ray && ln.tv.active));
        if (live0 == 0) { if (exhausted && !__any(busy)) break; else continue; }
        const int leave_at = live0 > RT_MARCH_REFILL ? live0 - RT_MARCH_REFILL : 0;
#pragma unroll 1
        do {
            trace_round<COUNT, ACCEL, EXT, RT_MARCH_STACK, true, RT_PIPE_TRACE_DSTEPS>(ln.tv, busy && ln.has_ray, sc, (uint2 RT_L *)lds_stack, (float RT_L *)lds_tm, RT_GPTR(uint2, job.spill), job.n_threads, gtid, tc);
        } while (__popcll(__ballot(busy && ln.has_ray && ln.tv.active)) > leave_at);
    }
    if (COUNT) {
        unsigned long long v[5] = {c_any, tc.nodes, tc.leaf_refs, tc.tris, tc.spills};
        const int idx[5] = {2, 3, 4, 5, 7};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0 && x) atomicAdd(job.counters + idx[k], x);
        }
    }
}

typedef void (*PipeMarchFn)(const DevScene *, const DevFrame *, const PipePool *, MarchJob);

}  // namespace rt
