// rt_render_kernel.h -- the persistent-thread megakernel: Scene::Render's sample loop (scene.cpp:42-84) with every lane running
// the state machine of rt_integrate.h and all lanes of a wave sharing ONE traversal loop.  Used for small, cache-resident scenes
// (Cornell: VALU-bound); large scenes go through the queue pipeline of rt_pipeline.h.  Instantiated per integrator in
// rt_mega_{w,d,p}.hip (separate translation units so that the library builds in parallel).
#pragma once
#include "rt_integrate.h"

namespace rt {

#ifndef RT_MEGA_CHUNK
#define RT_MEGA_CHUNK 64          // camera samples a megakernel wave takes from the global work counter at a time
#endif
static_assert(RT_MEGA_CHUNK >= 64, "one fresh chunk must cover a whole wave's fetch");
// a value every lane of the wave holds equally, moved to scalar registers
RT_DEV unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
    return (unsigned long long)hi << 32 | lo;
}


// first work item of band r of n (multiples of the chunk size, so that a chunk never straddles two bands; band n ends at `total`)
RT_DEV unsigned long long band_lo(unsigned long long total, unsigned r, unsigned n) {
    if (r >= n) return total;
    return (total / RT_MEGA_CHUNK) * r / n * RT_MEGA_CHUNK;
}

// ------------------------------------------------------------------------------------------ kernels
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 1
#endif
// the path integrator by vertex (rt_integrate.h advance_pass_byv) in the register-capped flavour of the kd-tree / grid kernels without a medium
#ifndef RT_MEGA_BYV
#define RT_MEGA_BYV 1
#endif


// waves per SIMD of the high-occupancy flavour: 4 = 128 VGPRs.  5 (96 VGPRs) was marginally faster at one point but its
// spill placement swings with every code change (measured 108 -> 153 ms on the 1 M-triangle path frame for the same
// algorithm); 4 is stable: 101 ms there, 138 ms on the 100 k soup.
#ifndef RT_HIGH_OCC_WAVES
#define RT_HIGH_OCC_WAVES 4
#endif
// Scene and frame descriptors are read through pointers (uniform addresses -> scalar loads on demand) instead of
// being passed by value: the by-value form pinned >100 SGPRs and spilled them.
// MINW = minimum waves per SIMD the register allocator must make room for: 1 = natural allocation (~160 VGPRs, 3 waves/SIMD,
// best when VALU-bound: tiny cache-resident scenes); RT_HIGH_OCC_WAVES = 4 caps at 128 VGPRs (some spills to scratch) for
// 4 waves/SIMD: +20 % on the memory-latency-bound 100k..1M-triangle scenes, -15 % on Cornell.
template <bool COUNT, int INTEG, int ACCEL, bool VOL, int MINW, bool EXT>
__global__ __launch_bounds__(RT_BLOCK, MINW) void render_kernel(const DevScene *__restrict__ scp,
                                                                       const DevFrame *__restrict__ frp) {
    __shared__ uint2 lds_stack[RT_STACK_LDS * RT_BLOCK];
    constexpr bool POOL = MINW < RT_HIGH_OCC_WAVES;                       // the pooled-leaf scratch (19 KB) is only carried by the kernels that use it
    constexpr int PN = POOL ? RT_BLOCK : 64;
    __shared__ unsigned long long pool_key[PN];
    __shared__ float4 pool_res[PN];
    __shared__ unsigned pool_head[PN];
    __shared__ float4 pool_ray[3 * PN];
    // second plane of the pair-form stack (trace_round).  Only the register-capped flavour (MINW >= RT_HIGH_OCC_WAVES: the one large trees
    // use) runs the pair form; the natural-allocation kernels keep the index form (C2's kernel sits 2 VGPRs below the 3-wave step)
    __shared__ float lds_tm_own[POOL ? 1 : RT_STACK_LDS * RT_BLOCK];
    float RT_L *lds_tm = (float RT_L *)lds_tm_own;
    const DevScene &sc = *scp;
    const DevFrame &fr = *frp;
    const unsigned wave0 = POOL ? (threadIdx.x & ~63u) : 0u;
    const PoolLds pool = {(unsigned long long RT_L *)pool_key + wave0, (float4 RT_L *)pool_res + wave0, (unsigned RT_L *)pool_head + wave0,
                          (float4 RT_L *)pool_ray + 3 * wave0};
    const unsigned gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    constexpr bool BYV = RT_MEGA_BYV != 0 && !POOL && INTEG == RT_INTEGRATOR_PATH && !VOL && !EXT && !COUNT;
    Lane ln;
    ln.vf = 0u;
    ln.stage = ST_FETCH; ln.has_ray = false; ln.fsp = 0; ln.tv.active = false; ln.tv.hit_prim = -1;
    ln.L = mk3(0.f); ln.thr = mk3(1.f); ln.alpha = 0.f; ln.depth = 0; ln.specular = false;
    TravCounters tc; tc.nodes = tc.leaf_refs = tc.tris = tc.spills = 0;
    RT_PFT(tc.c_desc = tc.c_leaf = tc.n_chunks = tc.n_pooled = tc.n_iter = 0;)
    unsigned c_cam = 0, c_closest = 0, c_any = 0, c_bad = 0;

#ifdef RT_PROFILE
    unsigned long long pf_shade = 0, pf_trav = 0, pf_outer = 0, pf_inner = 0, pf_rounds = 0, pf_act = 0, pf_rays = 0, pf_t0 = 0;
#define RT_PF(x) x
#else
#define RT_PF(x)
#endif
#ifdef RT_TAIL_PROBE
    const unsigned long long tp_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long tp_empty = 0, tp_taken = 0;
#endif
    unsigned long long w_next = 0, w_end = 0;                              // this wave's chunk of the sample list (wave-uniform)
    const unsigned n_bands = fr.xcd_bands ? 8u : 1u, home = fr.xcd_bands ? (blockIdx.x & 7u) : 0u;
    unsigned band_shift = 0;                                               // bands this wave has seen the end of

    // phase gating (rt_integrate.h, stage_in_phase): sweeps alternate between the two halves of the path state machine;
    // the first sweep is of the second kind (it contains the work fetch)
    int phase = (INTEG == RT_INTEGRATOR_PATH && fr.phase_sync && !BYV) ? 1 : -1;
    for (;;) {
        RT_PF(pf_t0 = __builtin_readcyclecounter(); ++pf_outer;)
        // ---- shade / regenerate: run every lane that is not waiting on a ray until it is (or is out of work)
        do {
            RT_PF(++pf_inner;)
            if constexpr (BYV) advance_pass_byv<COUNT, ACCEL, EXT>(sc, fr, ln, gtid, &c_closest, &c_any, &c_bad);
            else advance_pass<COUNT, INTEG, VOL, EXT>(sc, fr, ln, gtid, &c_closest, &c_any, &c_bad, phase);
            const unsigned long long want = phase == 0 ? 0ull : __ballot(!ln.has_ray && ln.stage == ST_FETCH);
            if (want) {                                                   // work fetch from the wave's private chunk of the sample list
                // One device-scope counter serves ~6-8 ns per atomic whoever asks (measured through the trace kernel's refill rate,
                // profiles/r02_scan_util3.jsonl), so a wave goes to it once per RT_MEGA_CHUNK samples, not once per fetch.
                const unsigned n_want = unsigned(__popcll(want));
                const unsigned long long have = w_end - w_next;
                unsigned long long fresh = 0, fresh_end = 0;
                if (have < n_want) {                                      // wave-uniform branch
                    const int leader = __ffsll((long long)want) - 1;
                    // XCD bands (round 4): the work list is cut into 8 contiguous bands with their own counters; a wave draws from the band of ITS XCD
                    // (workgroup i runs on XCD i mod 8) and moves on to the next band when that one is used up.  With one counter the 4096 resident waves
                    // work on the same ~9 scanlines at any time and all eight private L2s hold the same lines; with bands each L2 holds one band's.
#pragma unroll 1
                    while (band_shift < n_bands) {
                        const unsigned r = (home + band_shift) % n_bands;
                        const unsigned long long lo = band_lo(fr.total_work, r, n_bands), hi = band_lo(fr.total_work, r + 1u, n_bands);
                        unsigned long long base = 0;
                        if (lane == leader) base = atomicAdd(fr.work_counter + 8u * r, (unsigned long long)RT_MEGA_CHUNK);
                        base = uniform64(__shfl(base, leader));
                        if (lo + base < hi) { fresh = lo + base; fresh_end = fresh + RT_MEGA_CHUNK < hi ? fresh + RT_MEGA_CHUNK : hi; break; }
                        ++band_shift;
                    }
                }
                const bool none_left = band_shift >= n_bands;             // every band's counter has passed its end
#ifdef RT_TAIL_PROBE
                if (none_left && !tp_empty) tp_empty = __builtin_amdgcn_s_memrealtime();
#endif
                const unsigned long long rk = __popcll(want & ((1ull << lane) - 1ull));
                const unsigned long long w_mine = rk < have ? w_next + rk : fresh + (rk - have);
                const bool got = rk < have || w_mine < fresh_end;          // (a chunk clipped at its band's end serves fewer lanes: the others ask again)
                if (have < n_want) {
                    const unsigned long long took = n_want - have < fresh_end - fresh ? n_want - have : fresh_end - fresh;
                    w_next = fresh + took; w_end = fresh_end;
                } else w_next += n_want;
                if (!ln.has_ray && ln.stage == ST_FETCH && !got) { if (none_left) ln.stage = ST_EXIT; }
                else if (!ln.has_ray && ln.stage == ST_FETCH) {
                    const unsigned long long w = w_mine;
#ifdef RT_TAIL_PROBE
                    tp_taken += 1;
#endif
                    {
                        unsigned long long pixel; int s;
                        bool ok;
                        if (fr.mega_tile > 0) { unsigned px; tile_order_to_sample(fr, unsigned(w), px, s); pixel = px; ok = true; }      // (single shard)
                        else ok = work_to_sample(fr, w, pixel, s);
                        if (ok) {
                            Ray ray;
                            setup_sample(sc, fr, ln, pixel, s, ray);
                            ln.work = fr.mega_tile > 0 ? uint32_t(pixel * unsigned(fr.spp) + unsigned(s)) : uint32_t(w);
                            if (INTEG == RT_INTEG_DIRECT_WEIGHTED) ln.ord = fr.weighted_phase == 1 ? 0u : RT_GPTR(const unsigned, fr.wt_base)[ln.work];
                            ln.L = mk3(0.f); ln.thr = mk3(1.f); ln.alpha = 0.f; ln.depth = 0; ln.fsp = 0;
                            ln.specular = false;
                            if (COUNT) ++c_cam;
                            accel_begin<ACCEL>(ln.tv, sc, ray, false);
                            if (VOL) vol_store_ray(fr, 0, gtid, ray);
                            ln.has_ray = true; ln.stage = ST_VERTEX;
                            if (BYV) ln.vf = BV_B | BV_CUR_B;                 // the camera ray is the "continuation" that leads to the first vertex
                        }
                    }
                }
            }
        } while (__any(!ln.has_ray && stage_in_phase(ln.stage, phase)));
        if (phase >= 0) phase ^= 1;
        RT_PF({ unsigned long long t1 = __builtin_readcyclecounter(); pf_shade += t1 - pf_t0; pf_t0 = t1; pf_rays += __popcll(__ballot(ln.has_ray && ln.tv.active)); })
        if (!__any(ln.has_ray)) {
            if (!__any(ln.stage != ST_EXIT)) break;
            continue;                                                     // everybody waits for the other kind of sweep
        }
        // ---- extend: one shared traversal loop.  Leave it early when only a few lanes are still traversing AND some
        // lane could meanwhile shade / fetch (its traversal state stays in registers + LDS and resumes next round).
        for (;;) {
            if constexpr (BYV) {                                          // a lane whose ray has ended starts its vertex's next ray here, without a shading pass
                const bool next = ln.has_ray && !ln.tv.active && (ln.vf & (BV_QM | BV_QB)) != 0u;
                if (__any(next)) { if (next) byv_ray_advance<COUNT, ACCEL, EXT>(sc, ln, &c_closest, &c_any); }       // (always at once: a lane left waiting here counts as "idle" for the exit test below)
            }
            const bool act = ln.has_ray && ln.tv.active;
            const unsigned long long am = __ballot(act);
            if (!am) break;
            RT_PF(++pf_rounds; pf_act += __popcll(am);)
            if (fr.exit_thresh > 0 && __popcll(am) <= fr.exit_thresh && __any(!act && ln.stage != ST_EXIT)) break;
            if (POOL && fr.trav_mode == 3) accel_round_pooled<COUNT, ACCEL, EXT>(ln.tv, ln.has_ray, sc, (uint2 RT_L *)lds_stack, RT_GPTR(uint2, fr.spill), fr.n_threads, gtid, tc, pool);
            else trace_round<COUNT, ACCEL, EXT, RT_STACK_LDS, !POOL>(ln.tv, ln.has_ray, sc, (uint2 RT_L *)lds_stack, lds_tm, RT_GPTR(uint2, fr.spill), fr.n_threads, gtid, tc, fr.leaf_min);
        }
        if (ln.has_ray && !ln.tv.active && !(BYV && (ln.vf & (BV_QM | BV_QB)) != 0u)) ln.has_ray = false;
        RT_PF(pf_trav += __builtin_readcyclecounter() - pf_t0;)
    }
#ifdef RT_PROFILE
    if (lane == 0) {
        unsigned long long v[12] = {pf_shade, pf_trav, pf_outer, pf_inner, pf_rounds, pf_act, pf_rays, tc.c_desc, tc.c_leaf, tc.n_chunks, tc.n_pooled, tc.n_iter};
        for (int k = 0; k < 12; ++k) atomicAdd(fr.counters + 8 + k, v[k]);
    }
#endif

#ifdef RT_TAIL_PROBE
    {
        for (int off = 32; off > 0; off >>= 1) tp_taken += __shfl_down(tp_taken, off);
        if (lane == 0 && fr.probe) {
            unsigned long long *o = fr.probe + size_t(gtid >> 6) * 4u;
            o[0] = tp_start; o[1] = tp_empty; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = tp_taken;
        }
    }
#endif
    if (COUNT && !(INTEG == RT_INTEG_DIRECT_WEIGHTED && fr.weighted_phase != 3)) {       // (a weighted frame's count and survey passes are not Scene::Render's rays)
        unsigned long long v[8] = {c_cam, c_closest, c_any, tc.nodes, tc.leaf_refs, tc.tris, c_bad, tc.spills};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0 && x) atomicAdd(fr.counters + k, x);
        }
    }
}

typedef void (*RenderKernelFn)(const DevScene *, const DevFrame *);

}  // namespace rt
