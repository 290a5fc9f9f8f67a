// rt_pipe_vertex.h -- the queue pipeline's shade kernel for PathIntegrator::Li without a medium: ONE pass per path VERTEX.
//
// pipe_shade_kernel (rt_pipeline.h) suspends a path at every ray, as the recursion of the reference does: at a vertex the shadow
// ray of EstimateDirect (transport.cpp:152-156), its BSDF-sampled ray (:176-191) and the path's continuation ray (path.cpp:111-143)
// are three passes, three loads / stores of the slot's whole state and three trace launches.  But nothing a path does at a vertex
// depends on what those rays return: the sample values and every RandomFloat() of the vertex are consumed in an order that no ray
// result changes (without a medium Scene::Transmittance draws nothing), the continuation direction is
// sampled from the BSDF alone, and the two EstimateDirect terms only decide whether two already computed contributions count.
// So this kernel issues all (up to) three rays of a vertex at once and adds
//     Ld = 0; if (!occluded) Ld += pendS; if (hit the sampled emitter, front side) Ld += pendM; L += thr_old * (Ld * nLights)
// -- the reference's own sums in the reference's own order (transport.cpp:127,155,190; path.cpp:99-110) -- when the slot resumes.
//   * slot state 7 float4 planes instead of 10, loaded / stored once per vertex instead of three times (the vertex itself -- p, nn,
//     sn, wo, the BSDF sample values -- is dead once its rays are out);
//   * a third of the iterations (trace launches with their tails, shade passes over sparse slots);
//   * the three rays of a vertex start at the same point and sit in neighbouring lanes of the trace kernel: their ~30-node
//     descent to the leaf that holds the origin asks for the same lines (one request per step for the three).
// The arithmetic is rt_integrate.h's stage bodies, called in the reference's order; films are bit-identical to the megakernel's.
#pragma once
#include "rt_pipeline.h"

namespace rt {

#define RT_PV_VEC 7               // float4 planes of slot state
// ctl word: mode (2 bits) | flags | depth << 8
enum { PV_FETCH = 0, PV_RESUME = 1, PV_EXIT = 2 };
enum { PV_S = 4u, PV_M = 8u, PV_B = 16u, PV_ED = 32u, PV_ENDED = 64u, PV_SPECULAR = 128u };
// ray kinds of a slot: queue entry = slot | kind << 30; ray_d / hit planes are indexed by kind
enum { PV_RAY_S = 0, PV_RAY_M = 1, PV_RAY_B = 2 };

template <bool COUNT, bool EXT>
__global__ __launch_bounds__(RT_BLOCK) void pipe_vertex_kernel(const DevScene *__restrict__ scp, const DevFrame *__restrict__ frp,
                                                                const PipePool *__restrict__ plp, PipeLaunch pk) {
    constexpr int INTEG = RT_INTEGRATOR_PATH;
    const DevScene &sc = *scp;
    const DevFrame &fr = *frp;
    const PipePool &pl = *plp;
    const unsigned slot = pk.slot_base + blockIdx.x * RT_BLOCK + threadIdx.x;   // slot_base and n_slots are multiples of RT_BLOCK
    const int lane = threadIdx.x & 63;
    const size_t n = pl.n_slots;
    const int nLights = int(sc.n_lights);
    Lane ln;
    ln.v.p = ln.v.nn = ln.v.ng = ln.v.sn = ln.v.tn = ln.v.wo = mk3(0.f); ln.v.mat = 0; ln.v.light = -1;
    ln.li = ln.lj = 0; ln.cur_light = 0; ln.Ld = ln.Ld_light = ln.L_all = ln.pend = mk3(0.f); ln.bs1 = ln.bs2 = ln.bcs = 0.f;
    ln.fsp = 0; ln.has_ray = false; ln.tv.active = false; ln.tv.hit_prim = -1; ln.tv.any = false;
    float4 RT_G *st = RT_GPTR(float4, pl.state) + slot;
    const float4 a0 = st[0], a1 = st[n];
    const unsigned ctl = __float_as_uint(a1.z);
    const unsigned mode = ctl & 3u;
    if (!__syncthreads_or(mode != PV_EXIT)) return;
    ln.sample_index = __float_as_uint(a0.x); ln.work = __float_as_uint(a0.y); ln.image_x = a0.z; ln.image_y = a0.w;
    ln.dim_base = __float_as_uint(a1.x); ln.rng.ctr = __float_as_uint(a1.y); ln.alpha = a1.w;
    ln.rng.base = rng_base(ln.sample_index, fr.seed);
    ln.depth = int((ctl >> 8) & 255u); ln.specular = (ctl & PV_SPECULAR) != 0;
    ln.L = mk3(0.f); ln.thr = mk3(1.f);
    ln.stage = mode == PV_EXIT ? ST_EXIT : ST_FETCH;
    unsigned c_cam = 0, c_closest = 0, c_any = 0, c_bad = 0;
    __shared__ unsigned blk_cnt, blk_base;
    if (threadIdx.x == 0) blk_cnt = 0u;
    unsigned long long RT_G *ww = RT_GPTR(unsigned long long, pl.wave_work) + size_t(slot >> 6) * 2;
    unsigned long long w_next = ww[0], w_end = ww[1];

    // ---- resume: the rays of this slot's previous vertex have been traced
    if (mode == PV_RESUME) {
        // every load of the resume is issued before the first use (a streaming kernel at 3 waves per SIMD lives on loads in flight; the
        // planes a slot does not need this time cost bandwidth, not latency)
        const float4 a2 = st[2 * n], a3 = st[3 * n], a4 = st[4 * n], a5 = st[5 * n], a6 = st[6 * n];
        const float4 ro = RT_GPTR(const float4, pl.ray_o)[slot];
        const float4 hS = RT_GPTR(const float4, pl.hit)[size_t(PV_RAY_S) * n + slot], hM = RT_GPTR(const float4, pl.hit)[size_t(PV_RAY_M) * n + slot],
                     hB = RT_GPTR(const float4, pl.hit)[size_t(PV_RAY_B) * n + slot];
        const float4 rdM = RT_GPTR(const float4, pl.ray_d)[size_t(PV_RAY_M) * n + slot], rdB = RT_GPTR(const float4, pl.ray_d)[size_t(PV_RAY_B) * n + slot];
        ln.L = mk3(a2.x, a2.y, a2.z); ln.cur_light = __float_as_int(a2.w);
        ln.thr = mk3(a3.x, a3.y, a3.z);
        if (ctl & PV_ED) {                                              // the two halves of EstimateDirect, then path.cpp:99-110
            const V3 thr_old = mk3(a4.x, a4.y, a4.z);
            V3 Ld = mk3(0.f);                                           // transport.cpp:127
            if (ctl & PV_S) {
                if (COUNT) ++c_any;
                if (__float_as_int(hS.x) < 0) Ld = Ld + mk3(a5.x, a5.y, a5.z) * mk3(1.f);     // unoccluded; Transmittance = 1 (no medium)
            }
            if (ctl & PV_M) {
                if (COUNT) ++c_closest;
                const int prim = __float_as_int(hM.x);
                if (prim >= 0) {                                        // transport.cpp:180-190
                    ln.tv.o = mk3(ro.x, ro.y, ro.z); ln.tv.d = mk3(rdM.x, rdM.y, rdM.z); ln.tv.mint = ro.w; ln.tv.maxt = hM.y;
                    ln.tv.hit_prim = prim; ln.tv.b1 = hM.z; ln.tv.b2 = hM.w;
                    V3 nh; int light;
                    prim_normal_light<EXT>(sc, ln.tv, nh, light);
                    if (light == ln.cur_light && dot3(nh, -ln.tv.d) > 0) Ld = Ld + mk3(a6.x, a6.y, a6.z) * mk3(1.f);
                }
            }
            ln.L = ln.L + thr_old * (Ld * float(nLights));
        }
        if (ctl & PV_B) {                                               // the continuation (or camera) ray: next vertex
            ln.tv.o = mk3(ro.x, ro.y, ro.z); ln.tv.mint = ro.w; ln.tv.d = mk3(rdB.x, rdB.y, rdB.z); ln.tv.maxt = rdB.w; ln.tv.any = false;
            ln.tv.hit_prim = __float_as_int(hB.x);
            if (ln.tv.hit_prim >= 0) ln.tv.maxt = hB.y;                 // primitive.cpp:120
            ln.tv.b1 = hB.z; ln.tv.b2 = hB.w;
            ln.stage = ST_VERTEX;
        } else ln.stage = ST_RETURN;                                    // the path ended at the previous vertex (PV_ENDED)
    }

    // ---- this vertex: everything up to the point where the reference would wait for a ray, for all of its rays
    unsigned flags = 0;
    bool waiting = false;                                               // this slot's vertex is done: its rays are recorded
    V3 thr_old = mk3(0.f), pendS = mk3(0.f), pendM = mk3(0.f);
    V3 ray_o = mk3(0.f); float ray_mint = 0.f;
    float4 dS = make_float4(0.f, 0.f, 0.f, 0.f), dM = dS, dB = dS;
#define RT_PV_BODY(S) stage_body<COUNT, INTEG, false, EXT, S, true>(sc, fr, ln, slot, &c_closest, &c_any, &c_bad)
    do {
        if (!waiting && ln.stage == ST_VERTEX) RT_PV_BODY(ST_VERTEX);               // -> ST_DIRECT_NEXT, or ST_RETURN (the ray left the scene)
        if (!waiting && ln.stage == ST_DIRECT_NEXT) {                           // UniformSampleOneLight -> EstimateDirect (transport.cpp:51-70,123-194)
            RT_PV_BODY(ST_DIRECT_NEXT);                                 // light-sampling half: shadow ray in ln.tv, or ST_ED_BSDF, or (no lights) ST_BOUNCE
            if (ln.has_ray) {
                ln.has_ray = false; ln.stage = ST_ED_BSDF; flags |= PV_S; pendS = ln.pend;
                ray_o = ln.tv.o; ray_mint = ln.tv.mint; dS = make_float4(ln.tv.d.x, ln.tv.d.y, ln.tv.d.z, ln.tv.maxt);
            }
            if (ln.stage == ST_ED_BSDF) {
                RT_PV_BODY(ST_ED_BSDF);                                 // BSDF-sampling half: MIS ray in ln.tv, or ST_ED_DONE
                if (ln.has_ray) {
                    ln.has_ray = false; flags |= PV_M; pendM = ln.pend;
                    ray_o = ln.tv.o; ray_mint = ln.tv.mint; dM = make_float4(ln.tv.d.x, ln.tv.d.y, ln.tv.d.z, ln.tv.maxt);
                }
                if (flags & (PV_S | PV_M)) { flags |= PV_ED; thr_old = ln.thr; }     // L += thr * (Ld * nLights) waits for the rays
                else { ln.stage = ST_ED_DONE; RT_PV_BODY(ST_ED_DONE); }              // Ld = 0: nothing to wait for
                ln.stage = ST_BOUNCE;
            }
        }
        if (!waiting && ln.stage == ST_BOUNCE) {                        // path.cpp:111-143
            RT_PV_BODY(ST_BOUNCE);
            if (ln.has_ray) {
                ln.has_ray = false; flags |= PV_B;
                ray_o = ln.tv.o; ray_mint = ln.tv.mint; dB = make_float4(ln.tv.d.x, ln.tv.d.y, ln.tv.d.z, ln.tv.maxt);
            }
        }
        if (!waiting && !(flags & PV_B) && ln.stage == ST_RETURN) {
            if (flags & PV_ED) flags |= PV_ENDED;                       // L is complete once the vertex's rays are back
            else { RT_PV_BODY(ST_RETURN); RT_PV_BODY(ST_POP); RT_PV_BODY(ST_FINISH); }   // -> ST_FETCH
        }
        const unsigned long long want = __ballot(flags == 0 && ln.stage == ST_FETCH);
        if (want) {                                                     // work fetch from the wave's chunk (as pipe_shade_kernel)
            const unsigned n_want = unsigned(__popcll(want));
            const unsigned long long have = w_end - w_next;
            unsigned long long fresh = 0;
            if (have < n_want) {                                        // wave-uniform branch
                if (lane == 0) fresh = atomicAdd(fr.work_counter, (unsigned long long)RT_WORK_CHUNK);
                fresh = __shfl(fresh, 0);
            }
            if (flags == 0 && ln.stage == ST_FETCH) {
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
                        flags = PV_B; ln.stage = ST_VERTEX;
                        ray_o = ray.o; ray_mint = ray.mint; dB = make_float4(ray.d.x, ray.d.y, ray.d.z, ray.maxt);
                    }
                }
            }
            if (have < n_want) { w_next = fresh + (n_want - have); w_end = fresh + RT_WORK_CHUNK; }
            else w_next += n_want;
        }
        waiting = (flags & (PV_S | PV_M | PV_B)) != 0;
    } while (__any(!waiting && ln.stage != ST_EXIT));
#undef RT_PV_BODY
    if (lane == 0) { ww[0] = w_next; ww[1] = w_end; }

    // ---- enqueue: the rays of a slot are neighbours in the queue (wave scan of the per-lane counts, one LDS atomic per wave, one global per workgroup)
    {
        unsigned RT_G *qc = RT_GPTR(unsigned, pl.q_count) + size_t(pk.qi) * RT_QC_STRIDE;
        const unsigned cnt = ((flags & PV_S) ? 1u : 0u) + ((flags & PV_M) ? 1u : 0u) + ((flags & PV_B) ? 1u : 0u);
        const unsigned incl = wave_scan_add(cnt);
        const unsigned wave_total = unsigned(__builtin_amdgcn_readlane(int(incl), 63));
        unsigned wbase = 0;
        __syncthreads();                                                // blk_cnt zeroed
        if (lane == 0 && wave_total) wbase = atomicAdd(&blk_cnt, wave_total);
        __syncthreads();
        if (threadIdx.x == 0 && blk_cnt) blk_base = atomicAdd((unsigned *)qc, blk_cnt);
        __syncthreads();
        unsigned q = pk.q_base + __shfl(wbase, 0) + blk_base + (incl - cnt);
        if (cnt) {
            unsigned RT_G *qe = RT_GPTR(unsigned, pl.q_slot);
            RT_GPTR(float4, pl.ray_o)[slot] = make_float4(ray_o.x, ray_o.y, ray_o.z, ray_mint);
            if (flags & PV_S) { qe[q++] = slot | (unsigned(PV_RAY_S) << 30); RT_GPTR(float4, pl.ray_d)[size_t(PV_RAY_S) * n + slot] = dS; }
            if (flags & PV_M) { qe[q++] = slot | (unsigned(PV_RAY_M) << 30); RT_GPTR(float4, pl.ray_d)[size_t(PV_RAY_M) * n + slot] = dM; }
            if (flags & PV_B) { qe[q++] = slot | (unsigned(PV_RAY_B) << 30); RT_GPTR(float4, pl.ray_d)[size_t(PV_RAY_B) * n + slot] = dB; }
        }
    }
    // ---- store the slot
    {
        const unsigned m = ln.stage == ST_EXIT ? unsigned(PV_EXIT) : unsigned(PV_RESUME);
        const unsigned c = m | flags | (ln.specular ? unsigned(PV_SPECULAR) : 0u) | (unsigned(ln.depth) << 8);
        st[0] = make_float4(__uint_as_float(ln.sample_index), __uint_as_float(ln.work), ln.image_x, ln.image_y);
        st[n] = make_float4(__uint_as_float(ln.dim_base), __uint_as_float(ln.rng.ctr), __uint_as_float(c), ln.alpha);
        if (m != PV_EXIT) {
            st[2 * n] = make_float4(ln.L.x, ln.L.y, ln.L.z, __int_as_float(ln.cur_light));
            st[3 * n] = make_float4(ln.thr.x, ln.thr.y, ln.thr.z, 0.f);
            if (flags & PV_ED) st[4 * n] = make_float4(thr_old.x, thr_old.y, thr_old.z, 0.f);
            if (flags & PV_S) st[5 * n] = make_float4(pendS.x, pendS.y, pendS.z, 0.f);
            if (flags & PV_M) st[6 * n] = make_float4(pendM.x, pendM.y, pendM.z, 0.f);
        }
    }
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

}  // namespace rt
