// rt_weighted.h -- DirectLighting with strategy "weighted": WeightedSampleOneLight (transport.cpp:71-122, chosen at directlighting.cpp:195-208).
//
// The reference picks the light of every shading point from a CDF of exponentially averaged reflected luminances (avgY[light], overallAvgY:
// members of the integrator) that EVERY shading point of the frame updates, in program order: sample after sample in the sampler's order, within a
// sample the specular recursion of directlighting.cpp:127-183 depth first (the point itself, its reflected subtree, its transmitted subtree).  The
// estimate at point k therefore depends on all k - 1 estimates before it.  What does NOT depend on the choices is the geometry of the frame: the
// camera rays, the specular tree under each of them, and the sample values EstimateDirect receives (one light sample, one BSDF sample, one light
// number per camera sample: directlighting.cpp:54-64) -- and, when every light draws the same number of RandomFloat()s per estimate (rt_render
// checks: 1 for an emitter of several triangles, ShapeSet::Sample shape.h:115-121, else 0), the RNG counter at every point.  So the frame runs as
//
//   pass 1 (megakernel, weighted_phase 1)  the specular trees alone: how many shading points each camera sample has           -> wt_base[work]
//          weighted_scan_*_kernel          exclusive scan: the ordinal of each sample's first point in program order (work = the sampler's order)
//   pass 2 (megakernel, weighted_phase 2)  at every point EstimateDirect for EVERY light with the point's sample values; kept: the light-number
//                                          sample and per light y(Ld) and y(nLights * Ld) -- all the recurrence ever looks at      -> wt_rec
//          weighted_recurrence_*_kernel    the recurrence itself, one wave, points in program order: light and lightSampleWeight     -> wt_pick
//   pass 3 (megakernel, weighted_phase 3)  Scene::Render proper with the light of every point read from wt_pick: samples, film, counters
//
// The estimator is the reference's, scale included: SampleStep1d's pdf is a density over [0, 1) (a light is chosen with probability pdf / nLights) and
// transport.cpp:118 divides by it, so a converged "weighted" image is the direct lighting / nLights.  Not corrected here: parity is the contract.
// Cost: the survey is an "all lights, one sample each" frame, so a weighted frame costs about (2 + nLights) frames of strategy "one" plus
// ~0.2 us per shading point of sequential arithmetic (four dependent IEEE divisions per point).  One shard only: the recurrence spans the frame.
#pragma once
#include "rt_device.h"

namespace rt {

// ---- exclusive scan of the per-sample point counts, three launches: per-block sums, the scan of those, the blocks' own scans ----------------------
#define RT_WSCAN_BLOCKS 256
#define RT_WSCAN_THREADS 256

__device__ __forceinline__ unsigned long long wscan_block_exclusive(unsigned long long v, unsigned long long *wave_tot, unsigned long long &block_total) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d); if (lane >= unsigned(d)) inc += o; }
    if (lane == 63u) wave_tot[wave] = inc;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    for (unsigned w = 0; w < RT_WSCAN_THREADS / 64; ++w) { const unsigned long long t = wave_tot[w]; if (w < wave) before += t; all += t; }
    __syncthreads();
    block_total = all;
    return before + inc - v;
}
__device__ __forceinline__ void wscan_range(unsigned long long n, unsigned long long &lo, unsigned long long &hi) {
    const unsigned long long tile = RT_WSCAN_THREADS * 4ull, tiles = (n + tile - 1) / tile, per = (tiles + RT_WSCAN_BLOCKS - 1) / RT_WSCAN_BLOCKS;
    lo = per * blockIdx.x * tile; lo = lo < n ? lo : n; hi = lo + per * tile; hi = hi < n ? hi : n;
}
// block b: the sum of its contiguous range of counts -> sums[b]
__global__ __launch_bounds__(RT_WSCAN_THREADS) void weighted_scan_sums_kernel(const unsigned *__restrict__ base, unsigned long long n, unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long wave_tot[RT_WSCAN_THREADS / 64];
    unsigned long long lo, hi; wscan_range(n, lo, hi);
    unsigned long long sum = 0;
    for (unsigned long long i = lo + threadIdx.x; i < hi; i += RT_WSCAN_THREADS) sum += base[i];
    unsigned long long total; wscan_block_exclusive(sum, wave_tot, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one block: sums[] -> exclusive; *total and base[n] = the frame's shading points
__global__ __launch_bounds__(RT_WSCAN_THREADS) void weighted_scan_top_kernel(unsigned *__restrict__ base, unsigned long long n, unsigned long long *__restrict__ sums, unsigned long long *__restrict__ total) {
    __shared__ unsigned long long wave_tot[RT_WSCAN_THREADS / 64];
    static_assert(RT_WSCAN_BLOCKS == RT_WSCAN_THREADS, "one sum per thread");
    unsigned long long all; const unsigned long long ex = wscan_block_exclusive(sums[threadIdx.x], wave_tot, all);
    sums[threadIdx.x] = ex;
    if (threadIdx.x == 0) { *total = all; base[n] = unsigned(all < 0xffffffffull ? all : 0xffffffffull); }      // (the host refuses a frame whose total does not fit 32 bits)
}
// block b: its range in place, counts -> ordinals, tiles of 4 consecutive counts per thread
__global__ __launch_bounds__(RT_WSCAN_THREADS) void weighted_scan_apply_kernel(unsigned *__restrict__ base, unsigned long long n, const unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long wave_tot[RT_WSCAN_THREADS / 64];
    unsigned long long lo, hi; wscan_range(n, lo, hi);
    unsigned long long carry = sums[blockIdx.x];
    for (unsigned long long t0 = lo; t0 < hi; t0 += RT_WSCAN_THREADS * 4ull) {
        const unsigned long long i0 = t0 + threadIdx.x * 4ull;
        unsigned c[4];
        for (int k = 0; k < 4; ++k) c[k] = i0 + k < hi ? base[i0 + k] : 0u;
        unsigned long long tile_total; unsigned long long run = carry + wscan_block_exclusive((unsigned long long)c[0] + c[1] + c[2] + c[3], wave_tot, tile_total);
        for (int k = 0; k < 4; ++k) { if (i0 + k < hi) base[i0 + k] = unsigned(run); run += c[k]; }
        carry += tile_total;
    }
}

// ---- the recurrence ------------------------------------------------------------------------------------------------------------------------
// One wave; its lanes stage `chunk` records at a time in LDS and write the picks back.  Dynamic LDS: [state (LDS form only) | chunk * (1 + 2 * nL)
// record floats | chunk float2 picks].
//
// Lane form (nL <= 64): light i lives in lane i.  Per point: avgYsample and its division by nLights in all lanes at once; the CDF's running sum in the
// reference's order (one scalar add per light: float addition is not associative); one division by c for every CDF entry; std::upper_bound as a
// ballot; the two divisions of SampleStep1d (offset along the segment, pdf) in two lanes of ONE division; t; the light.  Four dependent IEEE divisions
// per point instead of 2 * nL + 3 in a single lane -- a wave64 instruction costs its 4 cycles whether 1 or 64 lanes are active.
__device__ __forceinline__ float wt_readlane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

__global__ __launch_bounds__(64) void weighted_recurrence_lanes_kernel(const float *__restrict__ rec, float2 *__restrict__ pick, unsigned long long n_points, int nL, int chunk) {
    extern __shared__ float wt_lds[];
    const int R = 1 + 2 * nL, lane = threadIdx.x;
    float *buf = wt_lds;
    float2 *out = (float2 *)(buf + ((chunk * R + 1) & ~1));
    const bool mine = lane < nL;
    const float fn = float(nL);
    float avgY = 0.f, overall = 0.f;
    for (unsigned long long base = 0; base < n_points; base += unsigned(chunk)) {
        const int cnt = int(n_points - base < (unsigned long long)chunk ? n_points - base : (unsigned long long)chunk);
        const float *src = rec + base * (unsigned long long)R;
        for (int k = lane; k < cnt * R; k += 64) buf[k] = src[k];
        __syncthreads();
        float u_n = buf[0], ya_n = mine ? buf[1 + 2 * lane] : 0.f, yb_n = mine ? buf[2 + 2 * lane] : 0.f;
        for (int v = 0; v < cnt; ++v) {
            const float u = u_n, ya = ya_n, yb = yb_n;
            if (v + 1 < cnt) { const float *r = buf + (v + 1) * R; u_n = r[0]; if (mine) { ya_n = r[1 + 2 * lane]; yb_n = r[2 + 2 * lane]; } }   // next point's record: off the chain
            int light; float w;
            if (overall == 0.f) {                                                     // transport.cpp:87-98: UniformSampleOneLight seeds the table
                light = min(int(floorf(u * fn)), nL - 1);                             // transport.cpp:60-64
                const float lum = wt_readlane(yb, __builtin_amdgcn_readfirstlane(light));      // (nLights * Ld).y()
                overall = lum; avgY = lum; w = 0.f;
            } else {                                                                  // transport.cpp:99-119
                const float f = fmaxf(avgY, .1f * overall);                           // avgYsample[lane]
                const float q = f / fn;                                               // ComputeStep1dCDF mc.cpp:31-42: f[i - 1] / nSteps
                float run = 0.f, hi = 0.f;
                for (int k = 0; k < nL; ++k) { run = run + wt_readlane(q, k); if (lane == k) hi = run; }      // cdf[k + 1] = cdf[k] + f[k] / nSteps
                const float c = run;
                hi = hi / c;                                                          // cdf[lane + 1] /= c
                const unsigned long long above = __ballot(mine && u < hi);            // SampleStep1d mc.cpp:43-53: upper_bound = the first entry above u
                int off = above ? int(__builtin_ctzll(above)) : nL - 1;
                off = __builtin_amdgcn_readfirstlane(off);
                const float c_lo = off > 0 ? wt_readlane(hi, off - 1) : 0.f, c_hi = wt_readlane(hi, off), f_off = wt_readlane(f, off);      // cdf[offset] (cdf[0] stays 0), cdf[offset + 1], f[offset]
                const float num = lane == 0 ? u - c_lo : f_off, den = lane == 0 ? c_hi - c_lo : c;
                const float quo = num / den;                                          // lane 0: the offset along the segment, lane 1: *pdf = f[offset] / c
                const float uu = wt_readlane(quo, 0); w = wt_readlane(quo, 1);
                const float t = (float(off) + uu) / fn;
                light = __builtin_amdgcn_readfirstlane(min(int(fn * t), nL - 1));     // Float2Int(nLights * t) pbrt.h:614-621
                const float lum = wt_readlane(ya, light);
                if (lane == light) avgY = (1.f - .99f) * lum + .99f * avgY;           // ExponentialAverage pbrt.h:664-667
                overall = (1.f - .999f) * lum + .999f * overall;
            }
            if (lane == 0) out[v] = make_float2(__int_as_float(light), w);
        }
        __syncthreads();
        for (int k = lane; k < cnt; k += 64) pick[base + k] = out[k];
    }
}

// LDS form (64 < nL <= 2048): lane 0 walks the points with the integrator's three arrays in LDS.  r: the point's record.
__device__ __forceinline__ float2 weighted_point(const float *r, int n, float *avgY, float *f, float *cdf, float &overall) {
    const float u = r[0];
    int light; float w, lum;
    if (overall == 0.f) {
        light = min(int(floorf(u * float(n))), n - 1);
        lum = r[2 + 2 * light];
        overall = lum;
        for (int i = 0; i < n; ++i) avgY[i] = lum;
        w = 0.f;
    } else {
        for (int i = 0; i < n; ++i) f[i] = fmaxf(avgY[i], .1f * overall);
        cdf[0] = 0.f;
        for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + f[i - 1] / float(n);
        const float c = cdf[n];
        for (int i = 1; i < n + 1; ++i) cdf[i] /= c;
        int idx = 0;
        while (idx < n + 1 && !(u < cdf[idx])) ++idx;                                 // (cdf is non-decreasing: the linear and the binary search agree)
        idx = min(max(0, idx - 1), n - 1);
        const float uu = (u - cdf[idx]) / (cdf[idx + 1] - cdf[idx]);
        w = f[idx] / c;
        const float t = (float(idx) + uu) / float(n);
        light = min(int(float(n) * t), n - 1);
        lum = r[1 + 2 * light]; avgY[light] = (1.f - .99f) * lum + .99f * avgY[light];
        overall = (1.f - .999f) * lum + .999f * overall;
    }
    return make_float2(__int_as_float(light), w);
}
__global__ __launch_bounds__(64) void weighted_recurrence_lds_kernel(const float *__restrict__ rec, float2 *__restrict__ pick, unsigned long long n_points, int nL, int chunk) {
    extern __shared__ float wt_lds[];
    const int R = 1 + 2 * nL, n_state = (3 * nL + 1 + 1) & ~1;
    float *avgY = wt_lds, *f = wt_lds + nL, *cdf = wt_lds + 2 * nL;
    float *buf = wt_lds + n_state;
    float2 *out = (float2 *)(buf + ((chunk * R + 1) & ~1));
    float overall = 0.f;
    if (threadIdx.x == 0) for (int i = 0; i < nL; ++i) avgY[i] = 0.f;
    for (unsigned long long base = 0; base < n_points; base += unsigned(chunk)) {
        const int cnt = int(n_points - base < (unsigned long long)chunk ? n_points - base : (unsigned long long)chunk);
        const float *src = rec + base * (unsigned long long)R;
        for (int k = threadIdx.x; k < cnt * R; k += 64) buf[k] = src[k];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int v = 0; v < cnt; ++v) out[v] = weighted_point(buf + v * R, nL, avgY, f, cdf, overall);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 64) pick[base + k] = out[k];
    }
}


// ---- lights of mixed RNG use (DevFrame::wt_mixed) ------------------------------------------------------------------------------------------
// A sample with P shading points holds P * A + nD * P * (P + 1) record floats (A = 1 + 2 * (nL - nD); point j: A + 2 * nD * (j + 1)).
__global__ void weighted_sizes_kernel(const unsigned *__restrict__ base, unsigned long long n, unsigned A, unsigned nD, unsigned *__restrict__ recbase) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long P = base[i + 1] - base[i];
    const unsigned long long sz = P * A + (unsigned long long)nD * P * (P + 1);
    recbase[i] = sz > 0xffffffffull ? 0xffffffffu : unsigned(sz);        // (the scan's 64-bit total then passes 2^32 and the host refuses the frame)
}
// The recurrence for such a frame: as weighted_recurrence_lds_kernel (one lane walks, the integrator's arrays in LDS), sample by sample, with k = the
// drawing lights chosen so far in the sample: a drawing light's estimate is read from the k-th of its j + 1 survey entries.  Groups of consecutive
// samples are staged through LDS by all lanes; a sample whose records do not fit is read from HBM directly.
// LDS floats: avgY[nL] f[nL] cdf[nL + 1] ya[nL] yb[nL] | sb[group + 1] rb[group + 1] | (pad to even) buf[cap_floats] | out float2[cap_points]
__global__ __launch_bounds__(64) void weighted_recurrence_mixed_kernel(const float *__restrict__ rec, float2 *__restrict__ pick, const unsigned *__restrict__ base,
                                                                        const unsigned *__restrict__ recbase, unsigned long long n_samples, int nL, int nD,
                                                                        const unsigned *__restrict__ draws, int group, int cap_floats, int cap_points) {
    extern __shared__ float wt_lds[];
    float *avgY = wt_lds, *f = avgY + nL, *cdf = f + nL, *ya = cdf + nL + 1, *yb = ya + nL;
    unsigned *sb = (unsigned *)(yb + nL), *rb = sb + group + 1;
    float *buf = (float *)(rb + group + 1);
    buf += (buf - wt_lds) & 1;                                                 // an even float offset, so that the picks behind the records are 8-byte aligned
    float2 *out = (float2 *)(buf + ((cap_floats + 1) & ~1));
    const int lane = threadIdx.x;
    const unsigned A = 1u + 2u * unsigned(nL - nD);
    float overall = 0.f;
    if (lane == 0) for (int i = 0; i < nL; ++i) avgY[i] = 0.f;
    for (unsigned long long s0 = 0; s0 < n_samples;) {
        const int g = int(n_samples - s0 < (unsigned long long)group ? n_samples - s0 : (unsigned long long)group);
        for (int i = lane; i <= g; i += 64) { sb[i] = base[s0 + i]; rb[i] = recbase[s0 + i]; }
        __syncthreads();
        int take = 0;                                                          // samples of this round (wave-uniform: every lane reads the same LDS words)
        while (take < g && rb[take + 1] - rb[0] <= unsigned(cap_floats) && sb[take + 1] - sb[0] <= unsigned(cap_points)) ++take;
        const bool direct = take == 0;                                        // one sample larger than the staging buffers: straight from HBM
        if (direct) take = 1;
        if (!direct) for (unsigned i = lane; i < rb[take] - rb[0]; i += 64) buf[i] = rec[size_t(rb[0]) + i];
        __syncthreads();
        if (lane == 0) {
            for (int i = 0; i < take; ++i) {
                const unsigned P = sb[i + 1] - sb[i];
                const float *sr = direct ? rec + rb[i] : buf + (rb[i] - rb[0]);
                unsigned k = 0;
                for (unsigned j = 0; j < P; ++j) {
                    const float *r = sr + size_t(j) * A + size_t(nD) * j * (j + 1);
                    const float u = r[0];
                    unsigned off = 1;
                    for (int l = 0; l < nL; ++l) {
                        if (draws[l]) { ya[l] = r[off + 2 * k]; yb[l] = r[off + 2 * k + 1]; off += 2 * (j + 1); }
                        else { ya[l] = r[off]; yb[l] = r[off + 1]; off += 2; }
                    }
                    int light; float w, lum;
                    if (overall == 0.f) {                                      // transport.cpp:87-98
                        light = min(int(floorf(u * float(nL))), nL - 1);
                        lum = yb[light];
                        overall = lum;
                        for (int q = 0; q < nL; ++q) avgY[q] = lum;
                        w = 0.f;
                    } else {                                                   // transport.cpp:99-119, mc.cpp:31-53
                        for (int q = 0; q < nL; ++q) f[q] = fmaxf(avgY[q], .1f * overall);
                        cdf[0] = 0.f;
                        for (int q = 1; q < nL + 1; ++q) cdf[q] = cdf[q - 1] + f[q - 1] / float(nL);
                        const float c = cdf[nL];
                        for (int q = 1; q < nL + 1; ++q) cdf[q] /= c;
                        int idx = 0;
                        while (idx < nL + 1 && !(u < cdf[idx])) ++idx;
                        idx = min(max(0, idx - 1), nL - 1);
                        const float uu = (u - cdf[idx]) / (cdf[idx + 1] - cdf[idx]);
                        w = f[idx] / c;
                        const float t = (float(idx) + uu) / float(nL);
                        light = min(int(float(nL) * t), nL - 1);
                        lum = ya[light]; avgY[light] = (1.f - .99f) * lum + .99f * avgY[light];
                        overall = (1.f - .999f) * lum + .999f * overall;
                    }
                    if (draws[light]) ++k;                                      // the frame pass's counter moves on by this light's draw
                    const float2 pk = make_float2(__int_as_float(light), w);
                    if (direct) pick[size_t(sb[i]) + j] = pk; else out[sb[i] - sb[0] + j] = pk;
                }
            }
        }
        __syncthreads();
        if (!direct) for (unsigned i = lane; i < sb[take] - sb[0]; i += 64) pick[size_t(sb[0]) + i] = out[i];
        __syncthreads();
        s0 += unsigned(take);
    }
}

}  // namespace rt
