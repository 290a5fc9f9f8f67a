// rt_kernels.hip -- gfx950 kernels and the C ABI (include/pbrt_hip.h) of libpbrt_hip.so.
//
// Kernels
//   render_kernel   persistent-thread wavefront renderer: Scene::Render's sample loop (scene.cpp:42-84).
//                   Every lane runs the state machine of rt_integrate.h; all lanes of a wave share ONE
//                   kd-tree traversal loop (rt_traverse.h) whatever kind of ray they carry (camera,
//                   bounce, MIS closest-hit, shadow any-hit); finished lanes refill from a global work
//                   counter with one wave-aggregated atomic.  No MFMA: the work is pointer chasing and
//                   3-vector arithmetic, bounded by HBM/L2 latency and bandwidth, not by dense math.
//   trace_kernel    Scene::Intersect / IntersectP for caller-supplied rays (unit parity entry points).
//   camera_kernel   Sampler + Camera::GenerateRay only.
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (parity with the reference's non-FMA build).
#include "rt_integrate.h"
#include "rt_internal.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <cmath>

namespace rt {

// ------------------------------------------------------------------------------------------ kernels
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 1
#endif
// waves per SIMD of the high-occupancy flavour: 4 = 128 VGPRs.  5 (96 VGPRs) was marginally faster at one point but its
// spill placement swings with every code change (measured 108 -> 153 ms on the 1 M-triangle path frame for the same
// algorithm); 4 is stable: 101 ms there, 138 ms on the 100 k soup.
#ifndef RT_HIGH_OCC_WAVES
#define RT_HIGH_OCC_WAVES 4
#endif
#ifndef RT_EXIT_THRESH
#define RT_EXIT_THRESH 0
#endif
#ifndef RT_LOCKSTEP
#define RT_LOCKSTEP 1
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
    const DevScene &sc = *scp;
    const DevFrame &fr = *frp;
    const unsigned wave0 = POOL ? (threadIdx.x & ~63u) : 0u;
    const PoolLds pool = {(unsigned long long RT_L *)pool_key + wave0, (float4 RT_L *)pool_res + wave0, (unsigned RT_L *)pool_head + wave0,
                          (float4 RT_L *)pool_ray + 3 * wave0};
    const unsigned gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    Lane ln;
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
    // phase gating (rt_integrate.h, stage_in_phase): sweeps alternate between the two halves of the path state machine;
    // the first sweep is of the second kind (it contains the work fetch)
    int phase = (INTEG == RT_INTEGRATOR_PATH && fr.phase_sync) ? 1 : -1;
    for (;;) {
        RT_PF(pf_t0 = __builtin_readcyclecounter(); ++pf_outer;)
        // ---- shade / regenerate: run every lane that is not waiting on a ray until it is (or is out of work)
        do {
            RT_PF(++pf_inner;)
            advance_pass<COUNT, INTEG, VOL, EXT>(sc, fr, ln, gtid, &c_closest, &c_any, &c_bad, phase);
            const unsigned long long want = phase == 0 ? 0ull : __ballot(!ln.has_ray && ln.stage == ST_FETCH);
            if (want) {                                                   // wave-aggregated work fetch
                const int leader = __ffsll((long long)want) - 1;
                unsigned long long base = 0;
                if (lane == leader) base = atomicAdd(fr.work_counter, (unsigned long long)__popcll(want));
                base = __shfl(base, leader);
                if (!ln.has_ray && ln.stage == ST_FETCH) {
                    const unsigned long long w = base + __popcll(want & ((1ull << lane) - 1ull));
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
                            accel_begin<ACCEL>(ln.tv, sc, ray, false);
                            if (VOL) vol_store_ray(fr, 0, gtid, ray);
                            ln.has_ray = true; ln.stage = ST_VERTEX;
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
            const bool act = ln.has_ray && ln.tv.active;
            const unsigned long long am = __ballot(act);
            if (!am) break;
            RT_PF(++pf_rounds; pf_act += __popcll(am);)
            if (fr.exit_thresh > 0 && __popcll(am) <= fr.exit_thresh && __any(!act && ln.stage != ST_EXIT)) break;
            if (fr.trav_mode == 1) accel_round<COUNT, ACCEL, EXT>(ln.tv, ln.has_ray, sc, (uint2 RT_L *)lds_stack, RT_GPTR(uint2, fr.spill), fr.n_threads, gtid, tc);
            else if (fr.trav_mode == 2) accel_round_batched<COUNT, ACCEL, EXT>(ln.tv, ln.has_ray, sc, (uint2 RT_L *)lds_stack, RT_GPTR(uint2, fr.spill), fr.n_threads, gtid, tc);
            else if (POOL && fr.trav_mode == 3) accel_round_pooled<COUNT, ACCEL, EXT>(ln.tv, ln.has_ray, sc, (uint2 RT_L *)lds_stack, RT_GPTR(uint2, fr.spill), fr.n_threads, gtid, tc, pool);
            else if (act) accel_step<COUNT, ACCEL, EXT>(ln.tv, sc, (uint2 RT_L *)lds_stack, RT_GPTR(uint2, fr.spill), fr.n_threads, gtid, tc);
        }
        if (ln.has_ray && !ln.tv.active) ln.has_ray = false;
        RT_PF(pf_trav += __builtin_readcyclecounter() - pf_t0;)
    }
#ifdef RT_PROFILE
    if (lane == 0) {
        unsigned long long v[12] = {pf_shade, pf_trav, pf_outer, pf_inner, pf_rounds, pf_act, pf_rays, tc.c_desc, tc.c_leaf, tc.n_chunks, tc.n_pooled, tc.n_iter};
        for (int k = 0; k < 12; ++k) atomicAdd(fr.counters + 8 + k, v[k]);
    }
#endif

    if (COUNT) {
        unsigned long long v[8] = {c_cam, c_closest, c_any, tc.nodes, tc.leaf_refs, tc.tris, c_bad, tc.spills};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0 && x) atomicAdd(fr.counters + k, x);
        }
    }
}

// ImageFilm::AddSample (film/image.cpp:103-142) as a gather: one thread per film pixel visits, in the reference's
// sample order (sample-pixel rows, then columns, then sample-in-pixel), every sample of this shard whose filter
// footprint can contain the pixel, and accumulates w*L, w*alpha, w on top of what the film already holds.  The
// footprint test and the filter-table lookup are the reference's own expressions, evaluated per sample.
// A 16x16-pixel workgroup stages the sample records of one sample-pixel row (chunked by columns) in LDS, so each
// 32-byte record is fetched from HBM/L2 once per workgroup instead of once per pixel in its footprint (25x for the
// 2x2 Mitchell filter).  Column blocks are padded by one float4 so that the 16 lanes of a row, which read 16
// consecutive columns at the same sample slot, hit 16 different 16-byte LDS slots (conflict-free ds_read_b128).
__global__ __launch_bounds__(256) void film_gather_kernel(const DevFrame *__restrict__ frp, int rx, int ry, int cols_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) float4 lds_rec[];
    const DevFrame &fr = *frp;
    const int nbx = (fr.x_pixel_count + 15) / 16;
    const int bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
    const int lx = bx * 16 + (threadIdx.x & 15), ly = by * 16 + (threadIdx.x >> 4);
    const bool live = lx < fr.x_pixel_count && ly < fr.y_pixel_count;
    const int x = fr.x_pixel_start + lx, y = fr.y_pixel_start + ly;
    const size_t plane = size_t(fr.x_pixel_count) * fr.y_pixel_count, px = size_t(ly) * fr.x_pixel_count + lx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
    if (live) { a0 = fr.accum[px]; a1 = fr.accum[plane + px]; a2 = fr.accum[2 * plane + px]; a3 = fr.accum[3 * plane + px]; a4 = fr.accum[4 * plane + px]; }
    // sample pixels whose samples (imageX in [sx, sx+1]) can reach pixel x: |x - (sx + u - .5)| <= width
    const int sx0 = max(int(ceilf(x - fr.fxw - 0.5f)), fr.x_start), sx1 = min(int(floorf(x + fr.fxw + 0.5f)), fr.x_end - 1);
    const int sy0 = max(int(ceilf(y - fr.fyw - 0.5f)), fr.y_start), sy1 = min(int(floorf(y + fr.fyw + 0.5f)), fr.y_end - 1);
    const int ew = fr.x_end - fr.x_start;
    const unsigned long long per_tile = (unsigned long long)fr.tile_pixels * fr.spp;
    const int xlo = fr.x_pixel_start, xhi = fr.x_pixel_start + fr.x_pixel_count - 1;
    const int ylo = fr.y_pixel_start, yhi = fr.y_pixel_start + fr.y_pixel_count - 1;
    const int X0 = fr.x_pixel_start + bx * 16, Y0 = fr.y_pixel_start + by * 16;
    const int bsx0 = max(X0 - rx, fr.x_start), bsx1 = min(X0 + 15 + rx, fr.x_end - 1);
    const int bsy0 = max(Y0 - ry, fr.y_start), bsy1 = min(Y0 + 15 + ry, fr.y_end - 1);
    const int spp = fr.spp;
    const float inv_fxw = fr.inv_fxw, inv_fyw = fr.inv_fyw;
    const int col_stride = fr.spp * 2 + 1;                              // float4 units, +1 pad
    unsigned long long *colbase = reinterpret_cast<unsigned long long *>(lds_rec + size_t(cols_per_chunk) * col_stride);
    __shared__ float ftab[256];                                         // FILTER_TABLE_SIZE^2 (film/image.cpp:53-64)
    ftab[threadIdx.x] = RT_GPTR(const float, fr.filter_table)[threadIdx.x];
    for (int sy = bsy0; sy <= bsy1; ++sy)
        for (int cx = bsx0; cx <= bsx1; cx += cols_per_chunk) {
            const int ncols = min(cols_per_chunk, bsx1 - cx + 1);
            __syncthreads();
            // one thread per column resolves where that sample pixel's records live in this shard's buffer (64-bit tile
            // arithmetic once per column, not once per staged float4)
            bool mine_col = false;
            if (int(threadIdx.x) < ncols) {
                const unsigned long long pixel = (unsigned long long)(sy - fr.y_start) * ew + (cx + int(threadIdx.x) - fr.x_start);
                bool mine; unsigned long long base;
                if (fr.total_pixels <= 0xffffffffull) {             // 32-bit divisions (a 64-bit one is ~150 VALU instructions)
                    const unsigned p32 = unsigned(pixel), tile = p32 / unsigned(fr.tile_pixels), in_tile = p32 - tile * unsigned(fr.tile_pixels);
                    const unsigned lt = tile / unsigned(fr.shard_count);
                    mine = int(tile - lt * unsigned(fr.shard_count)) == fr.shard_index;
                    base = ((unsigned long long)lt * per_tile + (unsigned long long)in_tile * fr.spp) * 2;
                } else {
                    const unsigned long long tile = pixel / fr.tile_pixels;
                    mine = int(tile % fr.shard_count) == fr.shard_index;
                    base = ((tile / fr.shard_count) * per_tile + (pixel % fr.tile_pixels) * fr.spp) * 2;
                }
                colbase[threadIdx.x] = mine ? base : ~0ull;
                mine_col = mine;
            }
            if (!__syncthreads_or(mine_col)) continue;        // this shard owns no sample pixel of this row chunk (7 of 8 chunks at 8 ranks)
            const int per_col = fr.spp * 2;
            int c = int(threadIdx.x) / per_col, k = int(threadIdx.x) - c * per_col;
            for (; c < ncols;) {
                const unsigned long long base = colbase[c];
#ifdef RT_GATHER_NOSTAGE
                if (false) {
#else
                if (base != ~0ull) {
#endif
                    float4 q = RT_GPTR(const float4, fr.samples)[base + k];
                    if (k & 1) {
                        // the sample's pixel footprint (film/image.cpp:108-116) depends on the sample only: computed once here by the
                        // staging thread and packed as two int16 pairs into the record's spare words, not once per pixel under it
                        const float dImageX = q.x - 0.5f, dImageY = q.y - 0.5f;
                        const int x0 = max(int(ceilf(dImageX - fr.fxw)), xlo), x1 = min(int(floorf(dImageX + fr.fxw)), xhi);
                        const int y0 = max(int(ceilf(dImageY - fr.fyw)), ylo), y1 = min(int(floorf(dImageY + fr.fyw)), yhi);
                        q.z = __uint_as_float((unsigned(x0) & 0xffffu) | (unsigned(x1) << 16));
                        q.w = __uint_as_float((unsigned(y0) & 0xffffu) | (unsigned(y1) << 16));
                    }
                    lds_rec[c * col_stride + k] = q;
                }
                k += 256;
                while (k >= per_col) { k -= per_col; ++c; }
            }
            __syncthreads();
#ifdef RT_GATHER_NOACC
            continue;
#endif
            if (!live || sy < sy0 || sy > sy1) continue;
            for (int sx = max(cx, sx0); sx <= min(cx + ncols - 1, sx1); ++sx) {
                const int c = sx - cx;
                if (colbase[c] == ~0ull) continue;
                const float4 *rec = lds_rec + c * col_stride;
                // four samples per trip: their records, footprint tests and filter weights are independent (8 + 4 LDS reads in
                // flight); only the five accumulations keep the reference's sample order
                int s = 0;
                for (; s + 4 <= spp; s += 4, rec += 8) {
                    float4 L[4], q[4]; float wt[4]; bool in[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { L[u] = rec[2 * u]; q[u] = rec[2 * u + 1]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int bx_ = __float_as_int(q[u].z), by_ = __float_as_int(q[u].w);
                        const int x0 = int(short(bx_ & 0xffff)), x1 = bx_ >> 16, y0 = int(short(by_ & 0xffff)), y1 = by_ >> 16;
                        in[u] = !(x < x0 || x > x1 || y < y0 || y > y1);
                        const float dImageX = q[u].x - 0.5f, dImageY = q[u].y - 0.5f;
                        const float fx = fabsf((x - dImageX) * inv_fxw * 16), fy = fabsf((y - dImageY) * inv_fyw * 16);
                        const int ifx = min(int(floorf(fx)), 15), ify = min(int(floorf(fy)), 15);
                        wt[u] = ftab[(ify * 16 + ifx) & 255];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (in[u]) {
                            a0 += wt[u] * L[u].x; a1 += wt[u] * L[u].y; a2 += wt[u] * L[u].z;   // Spectrum::AddWeighted color.h:116-120
                            a3 += L[u].w * wt[u]; a4 += wt[u];
                        }
                }
                for (; s < spp; ++s, rec += 2) {
                    const float4 q = rec[1];
                    const int bx_ = __float_as_int(q.z), by_ = __float_as_int(q.w);
                    const int x0 = int(short(bx_ & 0xffff)), x1 = bx_ >> 16, y0 = int(short(by_ & 0xffff)), y1 = by_ >> 16;
                    if (x < x0 || x > x1 || y < y0 || y > y1) continue;
                    const float dImageX = q.x - 0.5f, dImageY = q.y - 0.5f;
                    const float fx = fabsf((x - dImageX) * inv_fxw * 16), fy = fabsf((y - dImageY) * inv_fyw * 16);
                    const int ifx = min(int(floorf(fx)), 15), ify = min(int(floorf(fy)), 15);
                    const float wt = ftab[ify * 16 + ifx];
                    const float4 L = rec[0];
                    a0 += wt * L.x; a1 += wt * L.y; a2 += wt * L.z;       // Spectrum::AddWeighted color.h:116-120
                    a3 += L.w * wt; a4 += wt;
                }
            }
        }
    if (live) {
        fr.accum[px] = a0; fr.accum[plane + px] = a1; fr.accum[2 * plane + px] = a2; fr.accum[3 * plane + px] = a3;
        fr.accum[4 * plane + px] = a4;
    }
}

// ImageFilm::WriteImage (film/image.cpp:157-203) on the device: XYZ round trip (color.h:177-184, color.cpp:35-43),
// divide by the weight sum, clamps, premultiply.  out = rgb[H][W][3] then alpha[H][W].
__global__ void film_resolve_kernel(const float *__restrict__ accum, size_t n, int premultiply, float *__restrict__ rgb,
                                    float *__restrict__ alpha) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c0 = accum[i], c1 = accum[n + i], c2 = accum[2 * n + i];
    float xyz0 = 0.f, xyz1 = 0.f, xyz2 = 0.f;
    xyz0 += 0.412453f * c0; xyz1 += 0.212671f * c0; xyz2 += 0.019334f * c0;
    xyz0 += 0.357580f * c1; xyz1 += 0.715160f * c1; xyz2 += 0.119193f * c1;
    xyz0 += 0.180423f * c2; xyz1 += 0.072169f * c2; xyz2 += 0.950227f * c2;
    float r = 3.240479f * xyz0 + -1.537150f * xyz1 + -0.498535f * xyz2;
    float g = -0.969256f * xyz0 + 1.875991f * xyz1 + 0.041556f * xyz2;
    float b = 0.055648f * xyz0 + -0.204043f * xyz1 + 1.057311f * xyz2;
    float a = accum[3 * n + i];
    const float ws = accum[4 * n + i];
    if (ws != 0.f) {
        const float inv = 1.f / ws;
        r = clampf(r * inv, 0.f, RT_INF); g = clampf(g * inv, 0.f, RT_INF); b = clampf(b * inv, 0.f, RT_INF);
        a = clampf(a * inv, 0.f, 1.f);
    }
    if (premultiply) { r *= a; g *= a; b *= a; }
    rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b; alpha[i] = a;
}

__global__ __launch_bounds__(RT_BLOCK) void trace_kernel(DevScene sc, const RtRay *rays, unsigned n, int any,
                                                         RtHit *hits, unsigned char *occ, uint2 *spill,
                                                         unsigned n_threads, unsigned long long *counters) {
    __shared__ uint2 lds_stack[RT_STACK_LDS * RT_BLOCK];
    const unsigned gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    TravCounters tc; tc.nodes = tc.leaf_refs = tc.tris = tc.spills = 0;
    for (unsigned i = gtid; i < n; i += n_threads) {
        Ray r; r.o = mk3(rays[i].o[0], rays[i].o[1], rays[i].o[2]); r.d = mk3(rays[i].d[0], rays[i].d[1], rays[i].d[2]);
        r.mint = rays[i].mint; r.maxt = rays[i].maxt;
        Trav tv;
        if (sc.accel_kind == RT_ACCEL_GRID) { grid_begin(tv, sc, r, any != 0); while (tv.active) grid_step<true, true>(tv, sc, tc); }
        else { trav_begin(tv, sc, r, any != 0); while (tv.active) trav_step<true, true>(tv, sc, (uint2 RT_L *)lds_stack, RT_GPTR(uint2, spill), n_threads, gtid, tc); }
        if (any) occ[i] = tv.hit_prim >= 0 ? 1 : 0;
        else { hits[i].prim = tv.hit_prim; hits[i].t = tv.hit_prim >= 0 ? tv.maxt : 0.f; hits[i].b1 = tv.b1; hits[i].b2 = tv.b2; }
    }
    if (counters) {
        atomicAdd(counters + 3, (unsigned long long)tc.nodes);
        atomicAdd(counters + 4, (unsigned long long)tc.leaf_refs);
        atomicAdd(counters + 5, (unsigned long long)tc.tris);
        atomicAdd(counters + 7, (unsigned long long)tc.spills);
    }
}

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
using namespace rt;

// render_kernel instantiations, index = ((VOL*2 + ACCEL)*2 + COUNT)*3 + INTEG; +24..35: the high-occupancy flavour
// (COUNT = false only), index = 24 + (VOL*2 + ACCEL)*3 + INTEG; +36..47: timed kernels with the glossy (plastic) lobes compiled
// in, index = 36 + (VOL*2 + ACCEL)*3 + INTEG.  The counting twins always carry the glossy code (they are not timed); the
// common timed kernels do not: powf and the second lobe cost ~17 VGPRs (path 150 -> 167, direct 176: one wave per SIMD less).
typedef void (*RenderKernelFn)(const DevScene *, const DevFrame *);
#define RT_K3(C, A, V, W, G) render_kernel<C, 0, A, V, W, G>, render_kernel<C, 1, A, V, W, G>, render_kernel<C, 2, A, V, W, G>
static const RenderKernelFn g_render_kernels[48] = {
    RT_K3(false, 0, false, RT_MIN_WAVES, false), RT_K3(true, 0, false, RT_MIN_WAVES, true), RT_K3(false, 1, false, RT_MIN_WAVES, false),
    RT_K3(true, 1, false, RT_MIN_WAVES, true),
    RT_K3(false, 0, true, RT_MIN_WAVES, false),  RT_K3(true, 0, true, RT_MIN_WAVES, true),  RT_K3(false, 1, true, RT_MIN_WAVES, false),
    RT_K3(true, 1, true, RT_MIN_WAVES, true),
    RT_K3(false, 0, false, RT_HIGH_OCC_WAVES, false), RT_K3(false, 1, false, RT_HIGH_OCC_WAVES, false), RT_K3(false, 0, true, RT_HIGH_OCC_WAVES, false),
    RT_K3(false, 1, true, RT_HIGH_OCC_WAVES, false),
    RT_K3(false, 0, false, RT_MIN_WAVES, true), RT_K3(false, 1, false, RT_MIN_WAVES, true), RT_K3(false, 0, true, RT_MIN_WAVES, true),
    RT_K3(false, 1, true, RT_MIN_WAVES, true)};
#undef RT_K3

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(RT_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

static void hip_warn(hipError_t e, const char *what) {
    if (e != hipSuccess) std::fprintf(stderr, "libpbrt_hip: %s failed: %s\n", what, hipGetErrorString(e));
}
#define HIPWARN(expr) hip_warn((expr), #expr)

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
    DevScene *dev_scene = nullptr; DevFrame *dev_frame = nullptr;   // descriptors in HBM (read with scalar loads)
    float4 *samples = nullptr; size_t samples_cap = 0;          // per-shard sample buffer
    float ms_render = 0.f, ms_gather = 0.f; hipEvent_t ev2 = nullptr;
    float *resolve_buf = nullptr; size_t resolve_cap = 0;
    float *vol_buf = nullptr; size_t vol_cap = 0;          // volume scratch: rays | state | samp
    RtVolume volume{};
    int spill_depth = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_timing = false;
    bool counting = true;
    uint32_t n_tris = 0;
};

// tri_frame() of rt_shade.h on the host: same operations in the same order (trianglemesh.cpp:248-274, shape.cpp:43-50,
// reflection.cpp:475-476)
static void host_tri_frame(const float *v, bool flip, float nn[3], float sn[3]) {
    const float du1 = 0.f - 1.f, du2 = 1.f - 1.f, dv1 = 0.f - 1.f, dv2 = 0.f - 1.f;
    const float determinant = du1 * dv2 - dv1 * du2;
    const float invdet = 1.f / determinant;
    float dpdu[3], dpdv[3];
    for (int a = 0; a < 3; ++a) {
        const float dp1 = v[a] - v[6 + a], dp2 = v[3 + a] - v[6 + a];
        dpdu[a] = invdet * ((dv2 * dp1) - (dv1 * dp2));
        dpdv[a] = invdet * ((-du2 * dp1) + (du1 * dp2));
    }
    float c[3] = {(dpdu[1] * dpdv[2]) - (dpdu[2] * dpdv[1]), (dpdu[2] * dpdv[0]) - (dpdu[0] * dpdv[2]), (dpdu[0] * dpdv[1]) - (dpdu[1] * dpdv[0])};
    float inv = 1.f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (int a = 0; a < 3; ++a) { nn[a] = c[a] * inv; if (flip) nn[a] = -1.f * nn[a]; }
    inv = 1.f / sqrtf(dpdu[0] * dpdu[0] + dpdu[1] * dpdu[1] + dpdu[2] * dpdu[2]);
    for (int a = 0; a < 3; ++a) sn[a] = dpdu[a] * inv;
}

template <class T>
static int upload(RtScene *s, const T *host, size_t n, const T **dev) {
    void *p = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    HIPCHK(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    if (n) HIPCHK(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<const T *>(p);
    return RT_OK;
}

static void fill_info(const KdTree &tree, const GridAccelData &g, int kind, uint32_t n_tris, RtAccelInfo *info);

extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(int *count) {
    if (!count) return fail(RT_EINVAL, "rt_device_count: null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RT_EDEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n; return RT_OK;
}

int rt_scene_create(const RtSceneDesc *d, int device, RtScene **out) {
    if (!d || !out) return fail(RT_EINVAL, "rt_scene_create: null argument");
    if (d->n_tris && (!d->tri_verts || !d->tri_material || !d->tri_light || !d->tri_flags))
        return fail(RT_EINVAL, "rt_scene_create: missing triangle arrays");
    if (d->accel.kind != RT_ACCEL_KDTREE && d->accel.kind != RT_ACCEL_GRID) return fail(RT_EINVAL, "rt_scene_create: unknown accelerator kind");
    for (uint32_t i = 0; i < d->n_tris; ++i)
        if (d->tri_material[i] >= d->n_materials) return fail(RT_EINVAL, "rt_scene_create: material index out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(RT_EDEVICE, "rt_scene_create: no HIP device visible (the product path has no CPU fallback)");
    RtScene *s = new RtScene();
    if (device >= 0) { hipError_t e = hipSetDevice(device); if (e != hipSuccess) { delete s; return fail(RT_EDEVICE, "hipSetDevice failed"); } }
    HIPCHK(hipGetDevice(&s->device));
    HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)); s->own_stream = true;
    HIPCHK(hipEventCreate(&s->ev0)); HIPCHK(hipEventCreate(&s->ev1));
    s->n_tris = d->n_tris;

    s->accel_kind = d->accel.kind;
    if (s->accel_kind == RT_ACCEL_GRID) {
        build_grid(d->tri_verts, d->n_tris, s->gridacc);
        s->tree.nodes = s->gridacc.voxels; s->tree.leaf_refs = s->gridacc.refs; s->tree.max_depth = 0;
        std::memcpy(s->tree.bounds, s->gridacc.bounds, sizeof s->tree.bounds); s->tree.build_seconds = s->gridacc.build_seconds;
    } else build_kdtree(d->tri_verts, d->n_tris, d->accel, s->tree);

    // triangles -> 48-byte records
    std::vector<DevTri> tris(d->n_tris);
    uint32_t n_quadric_slots = 0;
    for (uint32_t i = 0; i < d->n_tris; ++i) {
        const float *v = d->tri_verts + size_t(9) * i;
        uint32_t bits = uint32_t(d->tri_material[i]) | (uint32_t(d->tri_flags[i] & 1u) << 16);
        int32_t light = d->tri_light[i];
        float fb, fl; std::memcpy(&fb, &bits, 4); std::memcpy(&fl, &light, 4);
        if (d->tri_flags[i] & 2u) {                       // quadric slot: {index, -, -} | bits | light
            bits |= RT_PRIM_QUADRIC; std::memcpy(&fb, &bits, 4);
            float fi; std::memcpy(&fi, &n_quadric_slots, 4); ++n_quadric_slots;
            tris[i].q0 = make_float4(fi, 0.f, 0.f, 0.f); tris[i].q1 = make_float4(0.f, 0.f, 0.f, 0.f); tris[i].q2 = make_float4(0.f, fb, fl, 0.f);
            continue;
        }
        const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
        tris[i].q0 = make_float4(v[0], v[1], v[2], e1[0]);
        tris[i].q1 = make_float4(e1[1], e1[2], e2[0], e2[1]);
        tris[i].q2 = make_float4(e2[2], fb, fl, 0.f);
    }
    // per-triangle shading constants: tri_frame() (rt_shade.h) evaluated once on the host with the same float
    // expressions (this file is compiled -ffp-contract=off for the host too; sqrt and divide are IEEE on both sides)
    std::vector<float4> shade(size_t(2) * d->n_tris);
    if (n_quadric_slots != d->n_quadrics || (d->n_quadrics && !d->quadrics)) return fail(RT_EINVAL, "rt_scene_create: quadric slots do not match n_quadrics");
    s->has_ext = s->has_ext || d->n_quadrics > 0;
    for (uint32_t i = 0; i < d->n_tris; ++i) {
        float nn[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f};
        if (!(d->tri_flags[i] & 2u)) host_tri_frame(d->tri_verts + size_t(9) * i, (d->tri_flags[i] & 1u) != 0, nn, sn);
        uint32_t bits = uint32_t(d->tri_material[i]) | (uint32_t(d->tri_flags[i] & 1u) << 16) | ((d->tri_flags[i] & 2u) ? RT_PRIM_QUADRIC : 0u);
        int32_t light = d->tri_light[i];
        float fb, fl; std::memcpy(&fb, &bits, 4); std::memcpy(&fl, &light, 4);
        shade[2 * i] = make_float4(nn[0], nn[1], nn[2], fb);
        shade[2 * i + 1] = make_float4(sn[0], sn[1], sn[2], fl);
    }
    int rc;
    if ((rc = upload(s, shade.data(), shade.size(), &s->dev.tri_shade))) return rc;
    if ((rc = upload(s, tris.data(), tris.size(), &s->dev.tris))) return rc;
    {
        std::vector<DevQuadric> dq(d->n_quadrics);
        for (uint32_t i = 0; i < d->n_quadrics; ++i) {
            const RtQuadric &q = d->quadrics[i]; DevQuadric &o = dq[i];
            if (q.type < RT_QUADRIC_SPHERE || q.type > RT_QUADRIC_HYPERBOLOID) return fail(RT_EINVAL, "rt_scene_create: unknown quadric type");
            std::memcpy(o.w2o, q.world_to_object, sizeof o.w2o); std::memcpy(o.o2w, q.object_to_world, sizeof o.o2w);
            o.radius = q.radius; o.zmin = q.zmin; o.zmax = q.zmax; o.theta_min = q.theta_min; o.theta_max = q.theta_max; o.phi_max = q.phi_max;
            o.type = q.type; o.pad = 0;
            for (int c = 0; c < 3; ++c) { o.p1[c] = q.p1[c]; o.p2[c] = q.p2[c]; }
            o.a = q.a; o.c = q.c;
        }
        if ((rc = upload(s, dq.data(), dq.size(), &s->dev.quadrics))) return rc;
    }
    const uint2 *nodes_dev = nullptr;
    {   // one node of padding: the traversal may fetch node i+1 together with node i
        std::vector<uint2> padded(s->tree.nodes.size() + 1);
        std::memcpy(padded.data(), s->tree.nodes.data(), s->tree.nodes.size() * sizeof(uint2));
        padded.back() = make_uint2(3u, 0u);
        if ((rc = upload(s, padded.data(), padded.size(), &nodes_dev))) return rc;
    }
    s->dev.nodes = nodes_dev;
    if ((rc = upload(s, s->tree.leaf_refs.data(), s->tree.leaf_refs.size(), &s->dev.leaf_refs))) return rc;

    // materials (OrenNayar constants: reflection.h:268-277)
    std::vector<DevMaterial> mats(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const RtMaterial &m = d->materials[i]; DevMaterial &o = mats[i];
        o.type = m.type; o.ior = m.ior; o.on_a = 1.f; o.on_b = -1.f;
        for (int c = 0; c < 3; ++c) { o.r[c] = m.kd[c]; o.t[c] = m.kt[c]; }
        o.has_r = (m.kd[0] != 0.f || m.kd[1] != 0.f || m.kd[2] != 0.f);
        o.has_t = (m.kt[0] != 0.f || m.kt[1] != 0.f || m.kt[2] != 0.f);
        for (int c = 0; c < 3; ++c) o.ks[c] = m.ks[c];
        o.exponent = 0.f;
        for (int c = 0; c < 3; ++c) o.kr[c] = m.kr[c];
        o.has_g = (m.ks[0] != 0.f || m.ks[1] != 0.f || m.ks[2] != 0.f); o.has_kr = (m.kr[0] != 0.f || m.kr[1] != 0.f || m.kr[2] != 0.f);
        if (m.type == RT_MAT_PLASTIC || m.type == RT_MAT_UBER) { s->has_ext = true; float e = 1.f / m.roughness; if (e > 1000.f || std::isnan(e)) e = 1000.f; o.exponent = e; }
        if (m.type < RT_MAT_MATTE || m.type > RT_MAT_UBER) return fail(RT_EINVAL, "rt_scene_create: unknown material type");
        if (m.type == RT_MAT_MATTE && m.sigma != 0.f) {
            float sigma = (3.14159265358979323846f / 180.f) * m.sigma;
            float sigma2 = sigma * sigma;
            o.on_a = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
            o.on_b = 0.45f * sigma2 / (sigma2 + 0.09f);
        }
    }
    if ((rc = upload(s, mats.data(), mats.size(), &s->dev.materials))) return rc;

    // lights + emitter triangles with ShapeSet area CDF (shape.h:122-135)
    std::vector<float> ltris(size_t(d->n_light_tris) * 16, 0.f);
    std::vector<DevLight> lights(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        const RtLight &L = d->lights[i]; DevLight &o = lights[i];
        o.type = L.type; o.n_samples = L.n_samples < 1 ? 1 : L.n_samples;
        for (int c = 0; c < 3; ++c) { o.color[c] = L.color[c]; o.pos[c] = L.pos[c]; }
        o.first_tri = L.first_tri; o.n_tris = L.n_tris; o.reverse_orientation = L.reverse_orientation;
        o.flip_normal = L.flip_normal; o.area = 0.f;
        for (int c = 0; c < 3; ++c) o.dir[c] = L.dir[c];
        for (int c = 0; c < 9; ++c) o.w2l[c] = L.world_to_light[c];
        o.cos_total = L.cos_total_width; o.cos_falloff = L.cos_falloff_start;
        o.quadric = L.quadric_plus1 - 1;
        if (L.quadric_plus1 < 0 || uint32_t(L.quadric_plus1) > d->n_quadrics) return fail(RT_EINVAL, "rt_scene_create: light refers to a quadric out of range");
        if (L.type < RT_LIGHT_POINT || L.type > RT_LIGHT_DISTANT) return fail(RT_EINVAL, "rt_scene_create: unknown light type");
        if (L.type != RT_LIGHT_AREA) continue;
        if (size_t(L.first_tri) + L.n_tris > d->n_light_tris) return fail(RT_EINVAL, "rt_scene_create: light triangle range out of bounds");
        float area = 0.f; std::vector<float> areas;
        for (uint32_t k = 0; k < L.n_tris; ++k) {
            const float *v = d->light_tris + size_t(L.first_tri + k) * 9;
            float *q = &ltris[size_t(L.first_tri + k) * 16];
            std::memcpy(q, v, 9 * sizeof(float));
            { float nl[3], sn_unused[3]; host_tri_frame(v, L.flip_normal != 0, nl, sn_unused); q[12] = nl[0]; q[13] = nl[1]; q[14] = nl[2]; }
            // Triangle::Area trianglemesh.cpp:329-335
            float ax = v[3] - v[0], ay = v[4] - v[1], az = v[5] - v[2];
            float bx = v[6] - v[0], by = v[7] - v[1], bz = v[8] - v[2];
            float cx = (ay * bz) - (az * by), cy = (az * bx) - (ax * bz), cz = (ax * by) - (ay * bx);
            float a = 0.5f * sqrtf(cx * cx + cy * cy + cz * cz);
            q[9] = a; area += a; areas.push_back(a);
        }
        float prev = 0.f;
        for (uint32_t k = 0; k < L.n_tris; ++k) {
            float c = prev + areas[k] / area;
            ltris[size_t(L.first_tri + k) * 16 + 10] = c; prev = c;
        }
        o.area = (L.n_tris == 1) ? areas[0] : area;
    }
    if ((rc = upload(s, lights.data(), lights.size(), &s->dev.lights))) return rc;
    if ((rc = upload(s, ltris.data(), ltris.size(), &s->dev.light_tris))) return rc;

    s->dev.n_tris = d->n_tris; s->dev.n_lights = d->n_lights;
    s->dev.accel_kind = s->accel_kind;
    for (int a = 0; a < 3; ++a) { s->dev.nvox[a] = s->gridacc.nvox[a]; s->dev.gwidth[a] = s->gridacc.width[a]; s->dev.ginv_width[a] = s->gridacc.inv_width[a]; }
    std::memcpy(s->dev.bounds, s->tree.bounds, sizeof s->dev.bounds);
    s->dev.cam = d->camera; s->dev.vol = d->volume; s->volume = d->volume;

    // persistent launch geometry: as many resident blocks as the kernel's registers/LDS admit
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, s->device));
    {
        unsigned mx = 0;
        for (int k = 0; k < 48; ++k) {
            int per_cu = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)g_render_kernels[k], RT_BLOCK, 0));
            if (per_cu < 1) per_cu = 1;
            s->grids[k] = unsigned(prop.multiProcessorCount) * unsigned(per_cu);
            mx = s->grids[k] > mx ? s->grids[k] : mx;
        }
        s->grid = mx;
    }
    s->n_threads = s->grid * RT_BLOCK;
    s->spill_depth = s->tree.max_depth > RT_STACK_LDS ? s->tree.max_depth - RT_STACK_LDS + 1 : 1;
    HIPCHK(hipMalloc((void **)&s->spill, size_t(s->spill_depth) * s->n_threads * sizeof(uint2)));
    HIPCHK(hipMalloc((void **)&s->work_counter, sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&s->counters, 24 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(s->counters, 0, 24 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&s->filter_dev, 256 * sizeof(float)));
    HIPCHK(hipMalloc((void **)&s->dev_scene, sizeof(DevScene)));
    HIPCHK(hipMalloc((void **)&s->dev_frame, sizeof(DevFrame)));
    HIPCHK(hipMemcpy(s->dev_scene, &s->dev, sizeof(DevScene), hipMemcpyHostToDevice));
    HIPCHK(hipEventCreate(&s->ev2));
    *out = s;
    return RT_OK;
}

int rt_scene_destroy(RtScene *s) {
    if (!s) return RT_OK;
    HIPWARN(hipSetDevice(s->device));
    hipStreamSynchronize(s->stream);
    for (void *p : s->allocs) HIPWARN(hipFree(p));
    if (s->own_accum && s->accum) HIPWARN(hipFree(s->accum));
    HIPWARN(hipFree(s->spill)); HIPWARN(hipFree(s->work_counter)); HIPWARN(hipFree(s->counters)); HIPWARN(hipFree(s->filter_dev));
    if (s->frames) HIPWARN(hipFree(s->frames));
    if (s->samples) HIPWARN(hipFree(s->samples));
    if (s->resolve_buf) HIPWARN(hipFree(s->resolve_buf));
    if (s->vol_buf) HIPWARN(hipFree(s->vol_buf));
    HIPWARN(hipFree(s->dev_scene)); HIPWARN(hipFree(s->dev_frame));
    if (s->ev2) HIPWARN(hipEventDestroy(s->ev2));
    if (s->ev0) HIPWARN(hipEventDestroy(s->ev0));
    if (s->ev1) HIPWARN(hipEventDestroy(s->ev1));
    if (s->own_stream && s->stream) HIPWARN(hipStreamDestroy(s->stream));
    delete s;
    return RT_OK;
}

int rt_scene_set_stream(RtScene *s, void *hip_stream) {
    if (!s) return fail(RT_EINVAL, "null scene");
    if (s->own_stream && s->stream) { HIPWARN(hipStreamSynchronize(s->stream)); HIPWARN(hipStreamDestroy(s->stream)); }
    s->stream = static_cast<hipStream_t>(hip_stream); s->own_stream = false;
    return RT_OK;
}

int rt_scene_accel_info(const RtScene *s, RtAccelInfo *info) {
    if (!s || !info) return fail(RT_EINVAL, "null argument");
    fill_info(s->tree, s->gridacc, s->accel_kind, s->n_tris, info);
    return RT_OK;
}

int rt_scene_accel_copy(const RtScene *s, uint32_t *nodes, uint32_t *leaf_refs) {
    if (!s) return fail(RT_EINVAL, "null scene");
    if (nodes) std::memcpy(nodes, s->tree.nodes.data(), s->tree.nodes.size() * sizeof(Node));
    if (leaf_refs) std::memcpy(leaf_refs, s->tree.leaf_refs.data(), s->tree.leaf_refs.size() * sizeof(uint32_t));
    return RT_OK;
}

struct RtKdTree { KdTree tree; GridAccelData grid; int kind = RT_ACCEL_KDTREE; uint32_t n_tris = 0; };
static void fill_info(const KdTree &tree, const GridAccelData &g, int kind, uint32_t n_tris, RtAccelInfo *info) {
    info->n_nodes = uint32_t(tree.nodes.size()); info->n_leaf_refs = uint32_t(tree.leaf_refs.size());
    info->max_depth = uint32_t(tree.max_depth); info->n_tris = n_tris;
    std::memcpy(info->bounds, tree.bounds, sizeof info->bounds); info->build_seconds = tree.build_seconds;
    info->kind = kind;
    for (int a = 0; a < 3; ++a) {
        info->grid_nvoxels[a] = kind == RT_ACCEL_GRID ? g.nvox[a] : 0;
        info->grid_width[a] = kind == RT_ACCEL_GRID ? g.width[a] : 0.f;
        info->grid_inv_width[a] = kind == RT_ACCEL_GRID ? g.inv_width[a] : 0.f;
    }
}
int rt_accel_build(const float *tri_verts, uint32_t n_tris, const RtAccelParams *params, RtAccel **out) {
    if (!out || (n_tris && !tri_verts)) return fail(RT_EINVAL, "rt_accel_build: null argument");
    RtAccelParams p; std::memset(&p, 0, sizeof p);
    if (params) p = *params;
    RtKdTree *t = new RtKdTree(); t->n_tris = n_tris; t->kind = p.kind;
    if (p.kind == RT_ACCEL_GRID) {
        build_grid(tri_verts, n_tris, t->grid);
        t->tree.nodes = t->grid.voxels; t->tree.leaf_refs = t->grid.refs; t->tree.max_depth = 0;
        std::memcpy(t->tree.bounds, t->grid.bounds, sizeof t->tree.bounds); t->tree.build_seconds = t->grid.build_seconds;
    } else if (p.kind == RT_ACCEL_KDTREE) build_kdtree(tri_verts, n_tris, p, t->tree);
    else { delete t; return fail(RT_EINVAL, "rt_accel_build: unknown accelerator kind"); }
    *out = t; return RT_OK;
}
int rt_accel_info(const RtAccel *t, RtAccelInfo *info) {
    if (!t || !info) return fail(RT_EINVAL, "null argument");
    fill_info(t->tree, t->grid, t->kind, t->n_tris, info);
    return RT_OK;
}
int rt_accel_copy(const RtAccel *t, uint32_t *nodes, uint32_t *leaf_refs) {
    if (!t) return fail(RT_EINVAL, "null accelerator");
    if (nodes) std::memcpy(nodes, t->tree.nodes.data(), t->tree.nodes.size() * sizeof(Node));
    if (leaf_refs) std::memcpy(leaf_refs, t->tree.leaf_refs.data(), t->tree.leaf_refs.size() * sizeof(uint32_t));
    return RT_OK;
}
int rt_accel_destroy(RtAccel *t) { delete t; return RT_OK; }
int rt_kdtree_build(const float *tri_verts, uint32_t n_tris, const RtAccelParams *params, RtKdTree **out) {
    if (params && params->kind != RT_ACCEL_KDTREE) return fail(RT_EINVAL, "rt_kdtree_build: not a kd-tree description");
    return rt_accel_build(tri_verts, n_tris, params, out);
}
int rt_kdtree_info(const RtKdTree *t, RtAccelInfo *info) { return rt_accel_info(t, info); }
int rt_kdtree_copy(const RtKdTree *t, uint32_t *nodes, uint32_t *leaf_refs) { return rt_accel_copy(t, nodes, leaf_refs); }
int rt_kdtree_destroy(RtKdTree *t) { return rt_accel_destroy(t); }

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
    const unsigned long long n_tiles = (fr.total_pixels + fr.tile_pixels - 1) / fr.tile_pixels;
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
            n2.push_back(ns); n2.push_back(ns); n1.push_back(ns); }
    } else if (rd->integrator == RT_INTEGRATOR_DIRECT) { n2 = {1, 1}; n1 = {1, 1}; }
    else if (rd->integrator == RT_INTEGRATOR_PATH) { n1.assign(9, 1); n2.assign(9, 1); }
    else if (rd->integrator != RT_INTEGRATOR_WHITTED) return fail(RT_EINVAL, "unknown integrator");
    n1.push_back(1); n1.push_back(1);                   // the volume integrator's tau / scatter samples
    if (n1.size() > RT_MAX_DIM_REQ || n2.size() > RT_MAX_DIM_REQ) return fail(RT_EINVAL, "too many lights for the sample table");
    unsigned c = 0;
    fr.n1d = int(n1.size()); fr.n2d = int(n2.size());
    const unsigned P = unsigned(fr.spp);
    if (rd->sampler == RT_SAMPLER_STRATIFIED) {             // LatinHypercube: n*d floats then n*d shuffles per request
        for (size_t i = 0; i < n1.size(); ++i) { fr.one_d[i] = DimReq{c, c + unsigned(n1[i]), (unsigned short)n1[i], 1}; c += 2u * n1[i]; }
        for (size_t i = 0; i < n2.size(); ++i) { fr.two_d[i] = DimReq{c, c + 2u * n2[i], (unsigned short)n2[i], 2}; c += 4u * n2[i]; }
        fr.lhs_total = c;
        fr.pixgen_draws = fr.jitter ? 7u * P : 2u * P;      // stratified.cpp:99-117
    } else if (rd->sampler == RT_SAMPLER_RANDOM) {          // one float per value (random.cpp:107-112)
        for (size_t i = 0; i < n1.size(); ++i) { fr.one_d[i] = DimReq{c, 0, (unsigned short)n1[i], 1}; c += unsigned(n1[i]); }
        for (size_t i = 0; i < n2.size(); ++i) { fr.two_d[i] = DimReq{c, 0, (unsigned short)n2[i], 2}; c += 2u * n2[i]; }
        fr.lhs_total = c;
        fr.pixgen_draws = 5u * P;                           // random.cpp:88-92
    } else {                                                // per-pixel tables (lowdiscrepancy.cpp:93-104, sampling.h:152-174)
        c = (2 + 2 * P) + (2 + 2 * P) + (1 + 2 * P);        // image, lens, time blocks
        for (size_t i = 0; i < n1.size(); ++i) { fr.one_d[i] = DimReq{c, 0, (unsigned short)n1[i], 1}; c += 1u + unsigned(n1[i]) * P + P; }
        for (size_t i = 0; i < n2.size(); ++i) { fr.two_d[i] = DimReq{c, 0, (unsigned short)n2[i], 2}; c += 2u + unsigned(n2[i]) * P + P; }
        fr.lhs_total = 0;
        fr.pixgen_draws = c;
    }
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
        fr.trav_mode = tiny ? 3 : 2;                           // lock-step rounds with pooled leaf tests (C2: 83.0 vs 87.1 ms for plain
                                                               // lock-step); else batched rounds (measured best on 100k-1M triangle soups)
        // long divergent rays: let finished lanes refill early.  Tiny scenes: only with phase gating (path integrator), where 8
        // measured +1.5 % (16: -13 %); Whitted / DirectLighting on Cornell lose 10 % with any early exit
        fr.exit_thresh = tiny ? (rd->integrator == RT_INTEGRATOR_PATH ? 8 : 0) : 32;
        fr.high_occupancy = tiny ? 0 : 1;
        if (const char *e = std::getenv("PBRT_HIP_HIGH_OCC")) fr.high_occupancy = std::atoi(e);
        if (const char *e = std::getenv("PBRT_HIP_TRAV_MODE")) fr.trav_mode = std::atoi(e);
        if (const char *e = std::getenv("PBRT_HIP_EXIT_THRESH")) fr.exit_thresh = std::atoi(e);
        fr.dbg_x = fr.dbg_y = -1000000;
        if (const char *e = std::getenv("PBRT_HIP_DEBUG_PIXEL")) std::sscanf(e, "%d,%d", &fr.dbg_x, &fr.dbg_y);
        fr.phase_sync = tiny ? 1 : 0;                          // C2: 63.6 vs 82.4 ms; 100k/1M soups (early-exit rounds): 8 % slower
        if (const char *e = std::getenv("PBRT_HIP_PHASE_SYNC")) fr.phase_sync = std::atoi(e);
        if (fr.trav_mode == 3 && fr.high_occupancy) fr.trav_mode = 1;   // the high-occupancy kernels carry no pooled-leaf scratch
        if (fr.trav_mode < 0 || fr.trav_mode > 3) fr.trav_mode = 1;
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

static int trace_common(RtScene *s, const RtRay *rays, uint32_t n, int any, RtHit *hits, uint8_t *occ) {
    if (!s || !rays || (!hits && !occ)) return fail(RT_EINVAL, "null argument");
    HIPCHK(hipSetDevice(s->device));
    RtRay *drays = nullptr; void *dout = nullptr;
    const size_t out_bytes = any ? size_t(n) : size_t(n) * sizeof(RtHit);
    HIPCHK(hipMalloc((void **)&drays, size_t(n ? n : 1) * sizeof(RtRay)));
    HIPCHK(hipMalloc(&dout, out_bytes ? out_bytes : 1));
    HIPCHK(hipMemcpy(drays, rays, size_t(n) * sizeof(RtRay), hipMemcpyHostToDevice));
    HIPWARN(hipEventRecord(s->ev0, s->stream));
    hipLaunchKernelGGL(trace_kernel, dim3(s->grid), dim3(RT_BLOCK), 0, s->stream, s->dev, drays, n, any,
                       (RtHit *)(any ? nullptr : dout), (unsigned char *)(any ? dout : nullptr), s->spill, s->n_threads, s->counters);
    HIPWARN(hipEventRecord(s->ev1, s->stream)); HIPWARN(hipEventRecord(s->ev2, s->stream)); s->have_timing = true;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(any ? (void *)occ : (void *)hits, dout, out_bytes, hipMemcpyDeviceToHost));
    HIPWARN(hipFree(drays)); HIPWARN(hipFree(dout));
    return RT_OK;
}
int rt_trace_closest(RtScene *s, const RtRay *rays, uint32_t n, RtHit *hits_out) { return trace_common(s, rays, n, 0, hits_out, nullptr); }
int rt_trace_any(RtScene *s, const RtRay *rays, uint32_t n, uint8_t *occluded_out) { return trace_common(s, rays, n, 1, nullptr, occluded_out); }

int rt_film_bind(RtScene *s, void *device_accum, int32_t w, int32_t h) {
    if (!s || w < 1 || h < 1) return fail(RT_EINVAL, "rt_film_bind: bad argument");
    HIPCHK(hipSetDevice(s->device));
    if (s->own_accum && s->accum) { HIPWARN(hipFree(s->accum)); s->accum = nullptr; }
    s->film_w = w; s->film_h = h;
    if (device_accum) { s->accum = static_cast<float *>(device_accum); s->own_accum = false; }
    else {
        HIPCHK(hipMalloc((void **)&s->accum, size_t(5) * w * h * sizeof(float))); s->own_accum = true;
        HIPCHK(hipMemset(s->accum, 0, size_t(5) * w * h * sizeof(float)));
    }
    return RT_OK;
}
int rt_film_clear(RtScene *s) {
    if (!s || !s->accum) return fail(RT_ESTATE, "no film bound");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemsetAsync(s->accum, 0, size_t(5) * s->film_w * s->film_h * sizeof(float), s->stream));
    return RT_OK;
}
int rt_film_read(RtScene *s, float *host_accum) {
    if (!s || !s->accum || !host_accum) return fail(RT_ESTATE, "no film bound / null buffer");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(host_accum, s->accum, size_t(5) * s->film_w * s->film_h * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

// ImageFilm::WriteImage film/image.cpp:157-203; Spectrum::XYZ color.h:177-184, weights color.cpp:35-43
int rt_film_resolve(RtScene *s, int premultiply, float *rgb_out, float *alpha_out) {
    if (!s || !rgb_out || !alpha_out) return fail(RT_EINVAL, "null argument");
    if (!s->accum) return fail(RT_ESTATE, "no film bound");
    HIPCHK(hipSetDevice(s->device));
    const size_t n = size_t(s->film_w) * s->film_h;
    if (s->resolve_cap < n) {
        if (s->resolve_buf) HIPWARN(hipFree(s->resolve_buf));
        HIPCHK(hipMalloc((void **)&s->resolve_buf, n * 4 * sizeof(float))); s->resolve_cap = n;
    }
    hipLaunchKernelGGL(film_resolve_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s->stream, s->accum, n, premultiply,
                       s->resolve_buf, s->resolve_buf + 3 * n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(rgb_out, s->resolve_buf, 3 * n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(alpha_out, s->resolve_buf + 3 * n, n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return RT_OK;
}

int rt_render(RtScene *s, const RtRenderDesc *rd) {
    if (!s || !rd) return fail(RT_EINVAL, "null argument");
    HIPCHK(hipSetDevice(s->device));
    // recursion frames for whitted / directlighting
    if (rd->integrator != RT_INTEGRATOR_PATH) {
        const size_t need = size_t(rd->max_depth + 2) * RT_FRAME_WORDS * s->n_threads;
        if (need > s->frames_floats) {
            if (s->frames) { HIPCHK(hipStreamSynchronize(s->stream)); HIPWARN(hipFree(s->frames)); s->frames = nullptr; }
            HIPCHK(hipMalloc((void **)&s->frames, need * sizeof(float))); s->frames_floats = need;
        }
    }
    DevFrame fr; int rc = make_frame(s, rd, fr, true); if (rc) return rc;
    if (s->volume.present) {
        if (!(rd->step_size > 0.f)) return fail(RT_EINVAL, "rt_render: volume integrator stepsize must be positive");
        if (rd->volume_integrator != RT_VOLUME_EMISSION && rd->volume_integrator != RT_VOLUME_SINGLE) return fail(RT_EINVAL, "rt_render: unknown volume integrator");
        const float ex = s->volume.p1[0] - s->volume.p0[0], ey = s->volume.p1[1] - s->volume.p0[1], ez = s->volume.p1[2] - s->volume.p0[2];
        const double diag = std::sqrt(double(ex) * ex + double(ey) * ey + double(ez) * ez);
        const double nsteps = std::ceil(diag / rd->step_size) + 2;
        if (nsteps > 65536) return fail(RT_EINVAL, "rt_render: stepsize too small for the medium (more than 65536 march steps)");
        const int nmax = int(nsteps);
        const int levels = (rd->integrator == RT_INTEGRATOR_PATH) ? 1 : rd->max_depth + 2;
        const size_t samp_words = rd->volume_integrator == RT_VOLUME_SINGLE ? size_t(3) * nmax : 0;
        const size_t need = (size_t(levels) * 8 + 13 + samp_words) * s->n_threads;
        if (need > s->vol_cap) {
            if (s->vol_buf) { HIPCHK(hipStreamSynchronize(s->stream)); HIPWARN(hipFree(s->vol_buf)); s->vol_buf = nullptr; }
            HIPCHK(hipMalloc((void **)&s->vol_buf, need * sizeof(float))); s->vol_cap = need;
        }
        fr.vol_rays = s->vol_buf; fr.vol_state = s->vol_buf + size_t(levels) * 8 * s->n_threads;
        fr.vol_samp = fr.vol_state + size_t(13) * s->n_threads; fr.vol_nmax = nmax;
    }
    const bool skip_film = std::getenv("PBRT_HIP_DEBUG_NOFILM") != nullptr;   // perf experiments only
    if (fr.total_work > s->samples_cap) {
        if (s->samples) { HIPCHK(hipStreamSynchronize(s->stream)); HIPWARN(hipFree(s->samples)); s->samples = nullptr; }
        HIPCHK(hipMalloc((void **)&s->samples, size_t(fr.total_work) * 2 * sizeof(float4)));
        s->samples_cap = fr.total_work;
    }
    fr.samples = s->samples;
    int variant = (((s->volume.present ? 1 : 0) * 2 + (s->accel_kind == RT_ACCEL_GRID ? 1 : 0)) * 2 + (s->counting ? 1 : 0)) * 3 + rd->integrator;
    if (fr.high_occupancy && !s->counting) variant = 24 + ((s->volume.present ? 1 : 0) * 2 + (s->accel_kind == RT_ACCEL_GRID ? 1 : 0)) * 3 + rd->integrator;
    if (s->has_ext && !s->counting) variant = 36 + ((s->volume.present ? 1 : 0) * 2 + (s->accel_kind == RT_ACCEL_GRID ? 1 : 0)) * 3 + rd->integrator;
    HIPCHK(hipMemcpyAsync(s->filter_dev, rd->filter_table, 256 * sizeof(float), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->dev_frame, &fr, sizeof(DevFrame), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemsetAsync(s->work_counter, 0, sizeof(unsigned long long), s->stream));
    HIPCHK(hipEventRecord(s->ev0, s->stream));
    { hipError_t pre = hipGetLastError(); if (pre != hipSuccess) return fail(RT_EDEVICE, std::string("pending HIP error before launch: ") + hipGetErrorString(pre)); }
    if (s->grids[variant] == 0) return fail(RT_ESTATE, "render kernel variant has no resident grid");
    hipLaunchKernelGGL(g_render_kernels[variant], dim3(s->grids[variant]), dim3(RT_BLOCK), 0, s->stream,
                       (const DevScene *)s->dev_scene, (const DevFrame *)s->dev_frame);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev1, s->stream));
    if (!skip_film) {
        const unsigned gb = unsigned((fr.x_pixel_count + 15) / 16) * unsigned((fr.y_pixel_count + 15) / 16);
        const int grx = int(std::floor(fr.fxw + 0.5f)), gry = int(std::floor(fr.fyw + 0.5f));   // reach of a sample pixel: |x - sx| <= w + .5
        const size_t col_bytes = size_t(fr.spp * 2 + 1) * sizeof(float4);
        if (fr.x_pixel_start + fr.x_pixel_count > 32767 || fr.y_pixel_start + fr.y_pixel_count > 32767)
            return fail(RT_EINVAL, "rt_render: film coordinates beyond 32767 (the gather packs sample footprints as int16)");
        size_t lds_kb = 40;                                   // 3 workgroups per CU (measured 60 KB: 5.6 ms, 40 KB: 5.3 ms on C2)
        if (const char *e = std::getenv("PBRT_HIP_GATHER_LDS_KB")) lds_kb = size_t(std::max(4, std::atoi(e)));
        int cols = int((lds_kb << 10) / col_bytes);
        if (cols < 1) return fail(RT_EINVAL, "rt_render: more samples per pixel than the film gather stages in LDS (max ~1900)");
        if (cols > 16 + 2 * grx) cols = 16 + 2 * grx;
        const size_t lds_bytes = size_t(cols) * col_bytes + size_t(cols) * sizeof(unsigned long long) + 16;
        hipLaunchKernelGGL(film_gather_kernel, dim3(gb), dim3(256), lds_bytes, s->stream, s->dev_frame, grx, gry, cols);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(s->ev2, s->stream));
    s->have_timing = true;
#ifdef RT_PROFILE
    {   // tools/perf_sweep.py 'p': per-wave cycle split of the render kernel (debug builds only)
        unsigned long long v[24];
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(hipMemcpy(v, s->counters, sizeof v, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "RT_PROFILE shade_cyc=%llu trav_cyc=%llu outer=%llu inner=%llu rounds=%llu act_lane_rounds=%llu rays_at_trav_start=%llu desc_cyc=%llu leaf_cyc=%llu chunks=%llu pooled_rounds=%llu leaf_iters=%llu\n",
                     v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19]);
        HIPCHK(hipMemsetAsync(s->counters + 8, 0, 16 * sizeof(unsigned long long), s->stream));
#ifdef RT_PROFILE_STAGES
        unsigned long long st[64];
        HIPCHK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pf_stage), sizeof st));
        for (int k = 0; k < 32; ++k) if (st[2 * k + 1])
            std::fprintf(stderr, "RT_PROFILE_STAGE %d cyc=%llu passes=%llu lanes=%llu\n", k, st[2 * k], st[2 * k + 1] >> 40, st[2 * k + 1] & ((1ull << 40) - 1));
        std::memset(st, 0, sizeof st);
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_pf_stage), st, sizeof st));
#endif
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
    if (total_ms) *total_ms = t;        // render kernel + film gather
    if (kernel_ms) *kernel_ms = k;      // rt::render_kernel alone (the dominant kernel)
    return RT_OK;
}

}  // extern "C"
