// rt_kernels.hip -- gfx950 kernels and the C ABI (include/pbrt_hip.h) of libpbrt_hip.so.
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
#include "rt_render_kernel.h"
#include "rt_pipeline.h"
#include "rt_pipe_vertex.h"
#include "rt_pipe_march.h"
#include "rt_weighted.h"
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

// Which of this shard's local pixels (work index / spp, rt_integrate.h work_to_sample) is sample pixel (sx, sy) of the sample extent, and
// does this shard render it at all?  1-D tiles are tile_pixels consecutive scanline pixels dealt round-robin to the shards, 2-D tiles
// tile_w x tile_h blocks; the whole frame holds < 2^32 camera samples (make_frame), so 32-bit divisions do.
__device__ inline void gather_local_pixel(const DevFrame &fr, int sx, int sy, bool &mine, unsigned long long &lp) {
    const unsigned px = unsigned(sx - fr.x_start), py = unsigned(sy - fr.y_start);
    if (fr.tile_w > 0) {
        const unsigned tx = px / unsigned(fr.tile_w), ty = py / unsigned(fr.tile_h);
        const unsigned tile = ty * unsigned(fr.tiles_x) + tx, lt = tile / unsigned(fr.shard_count);
        const unsigned in_tile = (py - ty * unsigned(fr.tile_h)) * unsigned(fr.tile_w) + (px - tx * unsigned(fr.tile_w));
        mine = int(tile - lt * unsigned(fr.shard_count)) == fr.shard_index;
        lp = (unsigned long long)lt * unsigned(fr.tile_pixels) + in_tile;
    } else {
        const unsigned pixel = py * unsigned(fr.x_end - fr.x_start) + px;
        if (fr.shard_count == 1) { mine = true; lp = pixel; return; }
        const unsigned tile = pixel / unsigned(fr.tile_pixels), in_tile = pixel - tile * unsigned(fr.tile_pixels);
        const unsigned lt = tile / unsigned(fr.shard_count);
        mine = int(tile - lt * unsigned(fr.shard_count)) == fr.shard_index;
        lp = (unsigned long long)lt * unsigned(fr.tile_pixels) + in_tile;
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
                bool mine; unsigned long long base;
                gather_local_pixel(fr, cx + int(threadIdx.x), sy, mine, base);
                colbase[threadIdx.x] = mine ? base : ~0ull;
                mine_col = mine;
            }
            if (!__syncthreads_or(mine_col)) continue;        // this shard owns no sample pixel of this row chunk (7 of 8 chunks at 8 ranks)
            const int per_col = fr.spp * 2;
            int c = int(threadIdx.x) / per_col, k = int(threadIdx.x) - c * per_col;
            for (; c < ncols;) {
                const unsigned long long base = colbase[c];
                if (base != ~0ull) {
                    float4 q = RT_GPTR(const float4, fr.samples)[sample_slot(unsigned(base), unsigned(k) >> 1, spp) + (k & 1) * RT_SAMPLE_XY];
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

// ---- the film gather as a march down the image (round 3) ---------------------------------------------------------------------
// One lane per film-pixel COLUMN of a strip of `strip_rows` rows, 64 consecutive columns per wave.  The lane walks the sample rows
// that can reach its strip from top to bottom and keeps the accumulators of the (at most 2 ry + 1) pixel rows the current sample
// row can touch in registers, so a sample record is fetched once (one coalesced 1 KB read per wave, sample_slot() layout) and its
// x-footprint test and filter column index are computed once for all those rows; each row then costs its own y test, its filter
// row index, the table look-up and the five accumulations of ImageFilm::AddSample (film/image.cpp:103-142), in the reference's
// sample order (sample-pixel rows, columns, sample in pixel).  Every lane of the wave has the same live rows, so the loops are
// instantiated per live-row count K and nothing is computed for rows outside the strip; 31 % of the lanes of the staged kernel
// above did useful work (5 of a workgroup's 16 pixel rows per staged sample row), here all of them do.
// The footprint test of image.cpp:108-116, x0 = max(Ceil2Int(dImageX - xWidth), xPixelStart) <= x <= x1 = min(Floor2Int(dImageX +
// xWidth), xPixelStart + xPixelCount - 1), is evaluated for the integer film pixel x as (float)x >= dImageX - xWidth && (float)x <=
// dImageX + xWidth: x >= ceil(a) <=> x >= a and x <= floor(b) <=> x <= b for an integer x, and x lies inside the film anyway.
// A sample outside the pixel's footprint is accumulated with weight +0 instead of being skipped (no branch in the loop): x + (+-0) == x
// for every x but -0, and an accumulator never holds -0 -- it starts at +0 and round-to-nearest addition yields -0 only from (-0) + (-0);
// L is finite (sample_write zeroes NaN / infinite radiance as scene.cpp:60-74 does), so 0 * L is a zero.
struct MarchBatch { float4 L[4]; float2 q[4]; };
typedef float vfloat2 __attribute__((ext_vector_type(2)));

template <int K, int RYMAX>
__device__ __forceinline__ void march_row(const DevFrame &fr, const float RT_L *ftab, vfloat2 (&acc01)[2 * RYMAX + 1], vfloat2 (&acc23)[2 * RYMAX + 1],
                                          float (&acc4)[2 * RYMAX + 1], bool live, int x, int sy, int wy0, int rx) {
    const float xf = float(x), fxw = fr.fxw, fyw = fr.fyw, kx = fr.inv_fxw, ky = fr.inv_fyw;
    const int spp = fr.spp, nbatch = (spp + 3) >> 2, ncol = 2 * rx + 1;
    float yf[K];
#pragma unroll
    for (int i = 0; i < K; ++i) yf[i] = float(wy0 + i);
    const float4 RT_G *samples = RT_GPTR(const float4, fr.samples);
    // column j of the window: sample pixel (x - rx + j, sy); a lane whose column lies outside the sample extent or belongs to another shard
    // reads record 0 of the buffer and weighs it 0
    auto column = [&](int j, bool &act) __attribute__((always_inline)) -> const float4 RT_G * {
        const int sx = x - rx + j;
        act = live & (sx >= fr.x_start) & (sx < fr.x_end);
        unsigned long long lp = 0;
        if (act) { bool mine; gather_local_pixel(fr, sx, sy, mine, lp); act = mine; }
        return samples + (act ? sample_slot(unsigned(lp), 0u, spp) : 0ull);
    };
    auto load = [&](MarchBatch &b, const float4 RT_G *rec, int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int su = min(s0 + u, spp - 1);                            // a batch past the pixel's last sample re-reads it (weight 0)
            b.L[u] = rec[size_t(su) * 128];
            const float4 RT_G *qp = rec + size_t(su) * 128 + RT_SAMPLE_XY;
            b.q[u] = *(const float2 RT_G *)qp;
        }
    };
    // (v * inv_w) * 16 of image.cpp:124-132 as v * (inv_w * 16): scaling by 16 commutes with the rounding of the product (no overflow here; a
    // product small enough to underflow indexes entry 0 either way); Floor2Int of a non-negative value is the truncating conversion.
    // The five accumulations run as two packed-fp32 pairs and a scalar (v_pk_mul_f32 / v_pk_add_f32: IEEE per component, no contraction).
    const float kx16 = kx * 16, ky16 = ky * 16;
    auto eval = [&](const MarchBatch &b, bool act, int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const vfloat2 Lxy = {b.L[u].x, b.L[u].y}, Lzw = {b.L[u].z, b.L[u].w};
            const float dImageX = b.q[u].x - 0.5f, dImageY = b.q[u].y - 0.5f;
            const bool inx = act & (s0 + u < spp) & (xf >= dImageX - fxw) & (xf <= dImageX + fxw);
            const float ay = dImageY - fyw, by = dImageY + fyw;
            const int ifx4 = min(int(fabsf((xf - dImageX) * kx16)), 15) << 2;
            float wt[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int ify = min(int(fabsf((yf[i] - dImageY) * ky16)), 15);
                wt[i] = *(const float RT_L *)((const char RT_L *)ftab + ((ify << 6) + ifx4));
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float w = (inx & (yf[i] >= ay) & (yf[i] <= by)) ? wt[i] : 0.f;
                const vfloat2 w2 = {w, w};
                acc01[i] += w2 * Lxy; acc23[i] += Lzw * w2; acc4[i] += w;       // Spectrum::AddWeighted color.h:116-120, alpha, weight sum
            }
        }
    };
    // one loop over the (column, batch of 4 samples) pairs of the row, the next batch's records in flight while this one is evaluated
    MarchBatch cur, nxt;
    bool act_cur, act_nxt;
    const float4 RT_G *rec = column(0, act_cur);
    act_nxt = act_cur;
    load(cur, rec, 0);
    int j = 0, bi = 0;
    for (int n = ncol * nbatch; n > 0; --n) {
        const int s0 = bi * 4;
        int bn = bi + 1;
        if (bn == nbatch) { bn = 0; ++j; if (j < ncol) rec = column(j, act_nxt); }
        if (n > 1) load(nxt, rec, bn * 4);
        eval(cur, act_cur, s0);
        cur = nxt; act_cur = act_nxt; bi = bn;
    }
}

template <int RYMAX>
__global__ __launch_bounds__(64) void film_march_kernel(const DevFrame *__restrict__ frp, int rx, int ry, int strip_rows, int row0, int row_end) {
    constexpr int NR = 2 * RYMAX + 1;
    const DevFrame &fr = *frp;
    __shared__ float ftab_s[256];                                       // FILTER_TABLE_SIZE^2 (film/image.cpp:53-64)
    for (int i = threadIdx.x; i < 256; i += 64) ftab_s[i] = RT_GPTR(const float, fr.filter_table)[i];
    __syncthreads();
    const float RT_L *ftab = (const float RT_L *)ftab_s;
    const int nbx = (fr.x_pixel_count + 63) / 64;
    const int bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
    const int lx = bx * 64 + int(threadIdx.x);
    const bool live = lx < fr.x_pixel_count;
    const int x = fr.x_pixel_start + lx;
    const int ly0 = row0 + by * strip_rows, ly1 = min(ly0 + strip_rows, row_end) - 1;
    const int yabs0 = fr.y_pixel_start + ly0, yabs1 = fr.y_pixel_start + ly1;
    const size_t plane = size_t(fr.x_pixel_count) * fr.y_pixel_count;
    float RT_G *accum = RT_GPTR(float, fr.accum);
    vfloat2 acc01[NR], acc23[NR]; float acc4[NR];            // window row i: sum w*L.r, w*L.g | sum w*L.b, w*alpha | sum w
#pragma unroll
    for (int i = 0; i < NR; ++i) { acc01[i] = vfloat2{0.f, 0.f}; acc23[i] = vfloat2{0.f, 0.f}; acc4[i] = 0.f; }
    // rows wy0 .. wy0 + k - 1 of the strip are the ones sample row sy can reach: [max(sy - ry, yabs0), min(sy + ry, yabs1)]
    int wy0 = yabs0, k = 0;
    auto fetch = [&](int pos, int y) __attribute__((always_inline)) {            // bring pixel row y (what the film already holds) into window position pos
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (live) {
            const size_t px = size_t(y - fr.y_pixel_start) * fr.x_pixel_count + lx;
#pragma unroll
            for (int c = 0; c < 5; ++c) v[c] = accum[c * plane + px];
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {                       // selects: an `if (i == pos)` becomes a store through a phi of pointers and pins the window in scratch
            acc01[i].x = (i == pos) ? v[0] : acc01[i].x; acc01[i].y = (i == pos) ? v[1] : acc01[i].y;
            acc23[i].x = (i == pos) ? v[2] : acc23[i].x; acc23[i].y = (i == pos) ? v[3] : acc23[i].y;
            acc4[i] = (i == pos) ? v[4] : acc4[i];
        }
    };
    fetch(k++, yabs0);                           // the first sample row, yabs0 - ry, reaches row yabs0 only
    for (int sy = yabs0 - ry; sy <= yabs1 + ry; ++sy) {
        if (sy >= fr.y_start && sy < fr.y_end) {
            switch (k) {
            case 1: march_row<1, RYMAX>(fr, ftab, acc01, acc23, acc4, live, x, sy, wy0, rx); break;
            case 2: if (NR >= 2) march_row<(NR >= 2 ? 2 : 1), RYMAX>(fr, ftab, acc01, acc23, acc4, live, x, sy, wy0, rx); break;
            case 3: if (NR >= 3) march_row<(NR >= 3 ? 3 : 1), RYMAX>(fr, ftab, acc01, acc23, acc4, live, x, sy, wy0, rx); break;
            case 4: if (NR >= 4) march_row<(NR >= 4 ? 4 : 1), RYMAX>(fr, ftab, acc01, acc23, acc4, live, x, sy, wy0, rx); break;
            case 5: if (NR >= 5) march_row<(NR >= 5 ? 5 : 1), RYMAX>(fr, ftab, acc01, acc23, acc4, live, x, sy, wy0, rx); break;
            case 6: if (NR >= 6) march_row<(NR >= 6 ? 6 : 1), RYMAX>(fr, ftab, acc01, acc23, acc4, live, x, sy, wy0, rx); break;
            case 7: if (NR >= 7) march_row<(NR >= 7 ? 7 : 1), RYMAX>(fr, ftab, acc01, acc23, acc4, live, x, sy, wy0, rx); break;
            default: break;
            }
        }
        if (k > 0 && wy0 == sy - ry) {            // row wy0 is out of reach of the next sample row: it is complete
            if (live) {
                const size_t px = size_t(wy0 - fr.y_pixel_start) * fr.x_pixel_count + lx;
                accum[px] = acc01[0].x; accum[plane + px] = acc01[0].y; accum[2 * plane + px] = acc23[0].x; accum[3 * plane + px] = acc23[0].y;
                accum[4 * plane + px] = acc4[0];
            }
#pragma unroll
            for (int i = 0; i + 1 < NR; ++i) { acc01[i] = acc01[i + 1]; acc23[i] = acc23[i + 1]; acc4[i] = acc4[i + 1]; }
            ++wy0; --k;
        }
        if (sy + 1 + ry <= yabs1) fetch(k++, sy + 1 + ry);
    }
}

// ---- the film gather with one pixel per lane and the sample rows staged in LDS (round 3; the default for filters reaching 1 or 2 pixels) -----
// The march above is bound by HBM traffic: a lane re-reads every record once per column of its window (5x) and the rows of a strip's halo,
// 13 GB for the 2.1 GB of records of a 1024^2 x 64 spp frame.  Here a wave owns NC = 64 / (2 ry + 1) film-pixel columns of a strip and stages
// one sample row of the NC + 2 rx sample-pixel columns that reach them in LDS, each record read from HBM once per strip (x 1.33 for the column
// halo).  Lane (column xi, slot m) accumulates ONE pixel at a time: of the 2 ry + 1 pixel rows a sample row can reach, slot m takes the one
// whose row index is congruent to m, keeps it for the 2 ry + 1 consecutive sample rows that reach it, stores it and moves 2 ry + 1 rows down --
// every lane has exactly one pixel row to serve for every staged sample row.
// The staging lane evaluates, once per record, ImageFilm::AddSample's footprint test and filter-table index (film/image.cpp:108-132) for
// each of the 2 rx + 1 pixel columns and 2 ry + 1 pixel rows the sample can reach and packs them as 5-bit entries (inside << 4 | index) into
// two words next to the record; a pixel's weight is then one look-up in a 1024-entry table indexed by (y entry << 5 | x entry) that holds
// 0 wherever either "inside" bit is clear (see march_row for why a weight of +0 is the reference's "skip"), and its accumulation is two
// packed multiply-adds and an add.  Order per pixel: sample rows, then columns, then samples -- the reference's.
#ifndef RT_SLOT_UNROLL
#define RT_SLOT_UNROLL 8         // samples per trip of the accumulation pass: their LDS reads are issued together
#endif
#ifndef RT_SLOT_PF
#define RT_SLOT_PF 12            // lookahead for rows of more than 4 records per lane (238 VGPRs: two waves per SIMD, what a 64 spp row's LDS allows anyway)
#endif
// PF: records per lane of the NEXT sample row requested before the current row's accumulation pass (they arrive while it runs; 6 VGPRs each)
template <int RX, int RY, int PF>
__global__ __launch_bounds__(64, 2) void film_slot_kernel(const DevFrame *__restrict__ frp, int strip_rows, int row0, int row_end) {
    constexpr int NS = 2 * RY + 1, NC = 64 / NS, NCS = NC + 2 * RX;
    extern __shared__ __attribute__((aligned(16))) float4 slot_lds[];
    const DevFrame &fr = *frp;
    const int spp = fr.spp, lstride = spp + 1;                          // +1: consecutive columns fall on different LDS banks
    float4 RT_L *Larr = (float4 RT_L *)slot_lds;                        // [NCS][lstride] L.rgb, alpha
    uint2 RT_L *Warr = (uint2 RT_L *)(Larr + NCS * lstride);            // [NCS][lstride] x entries, y entries
    float RT_L *tab2 = (float RT_L *)(Warr + NCS * lstride);            // [1024]
    unsigned RT_L *colbase = (unsigned RT_L *)(tab2 + 1024);            // [2][NCS] local pixel of each staged column (this row | the next), ~0u: none
    const int l = int(threadIdx.x);
    for (int t = l; t < 1024; t += 64) {
        const bool in = ((t >> 9) & 1) & ((t >> 4) & 1);
        tab2[t] = in ? RT_GPTR(const float, fr.filter_table)[((t >> 5) & 15) * 16 + (t & 15)] : 0.f;
    }
    const int nbx = (fr.x_pixel_count + NC - 1) / NC;
    const int bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
    const int m = l / NC, xi = l - m * NC;
    const int lx = bx * NC + xi;
    const bool col_live = (m < NS) & (lx < fr.x_pixel_count);
    const int X0 = fr.x_pixel_start + bx * NC;                          // the strip's first pixel column; staged column ci is sample pixel X0 - RX + ci
    const int ly0 = row0 + by * strip_rows, ly1 = min(ly0 + strip_rows, row_end) - 1;   // film rows [row0, row_end): the whole film, or one band of it
    const int yabs0 = fr.y_pixel_start + ly0, yabs1 = fr.y_pixel_start + ly1;
    const size_t plane = size_t(fr.x_pixel_count) * fr.y_pixel_count;
    float RT_G *accum = RT_GPTR(float, fr.accum);
    const float4 RT_G *samples = RT_GPTR(const float4, fr.samples);
    const float fxw = fr.fxw, fyw = fr.fyw, kx16 = fr.inv_fxw * 16, ky16 = fr.inv_fyw * 16;   // (v * inv) * 16 == v * (inv * 16), see march_row
    const int nrec = NCS * spp;

    auto resolve = [&](int buf, int sy) __attribute__((always_inline)) -> bool {   // lanes 0 .. NCS-1: where the staged columns of sample row sy live
        unsigned base = ~0u;
        if (l < NCS) {
            const int sx = X0 - RX + l;
            if (sx >= fr.x_start && sx < fr.x_end) { bool mine; unsigned long long lp; gather_local_pixel(fr, sx, sy, mine, lp); if (mine) base = unsigned(lp); }
            colbase[buf * NCS + l] = base;
        }
        return base != ~0u;
    };
    // record r of a staged row: column r % NCS, sample r / NCS (a load instruction covers NCS consecutive columns of 64 / NCS samples)
    auto request = [&](int buf, int r, float4 &L, float2 &xy) __attribute__((always_inline)) -> bool {
        const int sv = r / NCS, ci = r - sv * NCS;
        bool ok = r < nrec;
        const unsigned base = ok ? colbase[buf * NCS + ci] : ~0u;
        ok = ok & (base != ~0u);
        const unsigned long long at = ok ? sample_slot(base, unsigned(sv), spp) : 0ull;
        L = samples[at]; xy = *(const float2 RT_G *)(samples + at + RT_SAMPLE_XY);
        return ok;
    };
    // the record's footprint tests and filter-table indices for every pixel column / row it can reach (film/image.cpp:108-132), into LDS
    auto put = [&](int r, int sy, bool ok, float4 L, const float2 &xy) __attribute__((always_inline)) {
        if (r >= nrec) return;
        const int sv = r / NCS, ci = r - sv * NCS;
        const float dImageX = xy.x - 0.5f, dImageY = xy.y - 0.5f;
        const float ax = dImageX - fxw, bx_ = dImageX + fxw, ay = dImageY - fyw, by_ = dImageY + fyw;
        unsigned wx = 0, wy = 0;
        {
#pragma unroll
            for (int p = 0; p <= 2 * RX; ++p) {
                const float xf = float(X0 - 2 * RX + ci + p);
                const unsigned e = (((xf >= ax) & (xf <= bx_)) ? 16u : 0u) | unsigned(min(int(fabsf((xf - dImageX) * kx16)), 15));
                wx |= e << (5 * p);
            }
#pragma unroll
            for (int p = 0; p <= 2 * RY; ++p) {
                const float yf = float(sy - RY + p);
                const unsigned e = (((yf >= ay) & (yf <= by_)) ? 16u : 0u) | unsigned(min(int(fabsf((yf - dImageY) * ky16)), 15));
                wy |= e << (5 * p);
            }
        }
        if (!ok) { wx = 0; wy = 0; L = make_float4(0.f, 0.f, 0.f, 0.f); }
        Larr[ci * lstride + sv] = L;
        Warr[ci * lstride + sv] = make_uint2(wx, wy);
    };

    vfloat2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f}; float a4 = 0.f;
    int q = m < NS ? m : 0;                                             // my pixel row is sy - RY + q
    float4 pL[PF]; float2 pxy[PF]; unsigned pok = 0;                    // the next row's first PF records per lane, requested a row ahead
    const int sy_lo = max(yabs0 - RY, fr.y_start), sy_hi = min(yabs1 + RY, fr.y_end - 1);
    int cur = 0;
    // 2-D shard tiles: a wave none of whose (at most a handful of) tiles belongs to this shard has nothing to add to its pixels
    if (fr.tile_w > 0 && fr.shard_count > 1) {
        if (sy_lo > sy_hi) return;
        const int cx0 = max(X0 - RX, fr.x_start), cx1 = min(X0 - RX + NCS - 1, fr.x_end - 1);
        bool any = false;
        if (cx0 <= cx1)
            for (unsigned ty = unsigned(sy_lo - fr.y_start) / unsigned(fr.tile_h); ty <= unsigned(sy_hi - fr.y_start) / unsigned(fr.tile_h); ++ty)
                for (unsigned tx = unsigned(cx0 - fr.x_start) / unsigned(fr.tile_w); tx <= unsigned(cx1 - fr.x_start) / unsigned(fr.tile_w); ++tx)
                    any |= int((ty * unsigned(fr.tiles_x) + tx) % unsigned(fr.shard_count)) == fr.shard_index;
        if (!any) return;
    }
    // a sample row none of whose staged columns belongs to this shard is neither staged nor accumulated (N ranks: N - 1 of N rows of a wave)
    bool cur_any = false;
    if (sy_lo <= sy_hi) {
        cur_any = __syncthreads_or(resolve(0, sy_lo));
        if (cur_any) {
#pragma unroll
            for (int k = 0; k < PF; ++k) pok |= (request(0, l + 64 * k, pL[k], pxy[k]) ? 1u : 0u) << k;
        }
    }
    for (int sy = yabs0 - RY; sy <= yabs1 + RY; ++sy) {
        const int y = sy - RY + q;
        const bool valid = col_live & (y >= yabs0) & (y <= yabs1);
        const size_t px = size_t(valid ? y - fr.y_pixel_start : 0) * fr.x_pixel_count + (valid ? lx : 0);
        if (q == NS - 1) {                                              // a new pixel: what the film already holds
            a01 = vfloat2{0.f, 0.f}; a23 = vfloat2{0.f, 0.f}; a4 = 0.f;
            if (valid) { a01.x = accum[px]; a01.y = accum[plane + px]; a23.x = accum[2 * plane + px]; a23.y = accum[3 * plane + px]; a4 = accum[4 * plane + px]; }
        }
        if (sy >= sy_lo && sy <= sy_hi) {
            if (cur_any) {
                __syncthreads();                                        // the previous row's accumulation pass is done with the staged row
#pragma unroll
                for (int k = 0; k < PF; ++k) put(l + 64 * k, sy, (pok >> k) & 1u, pL[k], pxy[k]);
                for (int r0 = l + 64 * PF; r0 < nrec; r0 += 256) {      // rows longer than the lookahead: four records per lane in flight
                    float4 L[4]; float2 xy[4]; bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ok[u] = request(cur, r0 + 64 * u, L[u], xy[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) put(r0 + 64 * u, sy, ok[u], L[u], xy[u]);
                }
            }
            const bool next_any = __syncthreads_or(sy < sy_hi ? resolve(cur ^ 1, sy + 1) : false);   // also: the staged row is complete
            if (next_any) {
                pok = 0;
#pragma unroll
                for (int k = 0; k < PF; ++k) pok |= (request(cur ^ 1, l + 64 * k, pL[k], pxy[k]) ? 1u : 0u) << k;
            }
            cur ^= 1;
            const bool row_any = cur_any;
            cur_any = next_any;
            const unsigned shy = valid ? unsigned(5 * q) : 25u;         // bits 25.. of the y word are clear: weight 0 for a lane without a pixel
            // the accumulation pass, column after column; RT_SLOT_UNROLL samples' LDS reads are issued together.  (A three-stage software pipeline --
            // records of batch b + 2 read, weights of b + 1 looked up, batch b accumulated -- measured slower: 2.36 vs 2.00 ms on C2.)
            if (row_any)
#pragma unroll
            for (int j = 0; j <= 2 * RX; ++j) {
                const float4 RT_L *Lp = Larr + (xi + j) * lstride;
                const uint2 RT_L *Wp = Warr + (xi + j) * lstride;
                const unsigned shx = unsigned(5 * (2 * RX - j));
                auto one = [&](const float4 &L, const uint2 &w) __attribute__((always_inline)) {
                    const unsigned t = (__builtin_amdgcn_ubfe(w.y, shy, 5u) << 5) | __builtin_amdgcn_ubfe(w.x, shx, 5u);
                    const float wt = tab2[t];
                    const vfloat2 w2 = {wt, wt}, Lxy = {L.x, L.y}, Lzw = {L.z, L.w};
                    a01 += w2 * Lxy; a23 += Lzw * w2; a4 += wt;        // Spectrum::AddWeighted color.h:116-120, alpha, weight sum
                };
                int s = 0;
                for (; s + RT_SLOT_UNROLL <= spp; s += RT_SLOT_UNROLL) {
                    float4 L[RT_SLOT_UNROLL]; uint2 w[RT_SLOT_UNROLL];
#pragma unroll
                    for (int u = 0; u < RT_SLOT_UNROLL; ++u) { L[u] = Lp[s + u]; w[u] = Wp[s + u]; }
#pragma unroll
                    for (int u = 0; u < RT_SLOT_UNROLL; ++u) one(L[u], w[u]);
                }
                for (; s < spp; ++s) one(Lp[s], Wp[s]);
            }
        }
        if (q == 0 && valid) {                                          // the last sample row that reaches my pixel is done
            accum[px] = a01.x; accum[plane + px] = a01.y; accum[2 * plane + px] = a23.x; accum[3 * plane + px] = a23.y; accum[4 * plane + px] = a4;
        }
        q = q == 0 ? NS - 1 : q - 1;
    }
}

// ImageFilm::WriteImage (film/image.cpp:157-203) on the device: XYZ round trip (color.h:177-184, color.cpp:35-43),
// divide by the weight sum, clamps, premultiply.  out = rgb[H][W][3] then alpha[H][W].
// `alpha` == nullptr: interleaved RGBA, out = rgba[n][4] (the payload of one all-gather, rt_film_resolve_device_rgba).
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
    if (alpha) { rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b; alpha[i] = a; }
    else reinterpret_cast<float4 *>(rgb)[i] = make_float4(r, g, b, a);
}

// N > 1 merge: a rank's full-frame film (5 planes of h x w) re-laid as `world` parts of `rows` film rows, part r = [5][rows][w] (rows beyond h: zero) --
// the send buffer of ONE reduce-scatter whose r-th chunk is everything rank r resolves (rt_film_pack_parts).  One float4 per thread where w allows.
__global__ void film_pack_parts_kernel(const float *__restrict__ accum, int w, int h, int rows, size_t n_out, float *__restrict__ parts) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;           // output index: ((part * 5 + plane) * rows + row) * w + x
    if (i >= n_out) return;
    const size_t x = i % size_t(w), t = i / size_t(w);
    const size_t row = t % size_t(rows), u = t / size_t(rows);
    const size_t plane = u % 5u, part = u / 5u;
    const size_t y = part * size_t(rows) + row;
    parts[i] = y < size_t(h) ? accum[(plane * size_t(h) + y) * size_t(w) + x] : 0.f;
}

// rt_samples_read: records [first, first + count) of the shard's work list, out of the sample_slot() layout, as 2 x float4 per sample
__global__ void samples_unpack_kernel(const float4 *__restrict__ samples, unsigned long long first, unsigned long long count, int spp, float4 *__restrict__ out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const unsigned long long w = first + i;
    const unsigned lp = unsigned(w / unsigned(spp));
    const unsigned long long at = sample_slot(lp, unsigned(w - (unsigned long long)lp * unsigned(spp)), spp);
    out[2 * i] = samples[at]; out[2 * i + 1] = samples[at + RT_SAMPLE_XY];
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

// Experiment / test knobs (PBRT_HIP_*: kernel flavour, pipeline form, film-gather kernel, layout switches, logs) are read only when
// PBRT_HIP_TUNE is set in the environment -- the tests and tools/ set it -- so a production process cannot change its behaviour through a stray
// variable; -DRT_NO_TUNABLES compiles them out.
static const char *knob(const char *name) {
#ifdef RT_NO_TUNABLES
    (void)name; return nullptr;
#else
    static const bool on = std::getenv("PBRT_HIP_TUNE") != nullptr;
    return on ? std::getenv(name) : nullptr;
#endif
}


// render_kernel instantiations live in rt_mega_{w,d,p}.hip, 16 per integrator: k = (VOL*2 + ACCEL)*2 + COUNT for the natural-allocation
// kernels (0..7; the counting twins always carry the glossy / quadric code, they are not timed), 8 + VOL*2 + ACCEL for the
// high-occupancy flavour, 12 + VOL*2 + ACCEL for the timed kernels with the glossy (plastic) lobes and quadric slots compiled in
// (EXT: powf and the second lobe cost ~17 VGPRs, one wave per SIMD less for DirectLighting).  `variant` keeps round 1's numbering:
// ((VOL*2 + ACCEL)*2 + COUNT)*3 + INTEG | 24 + (VOL*2 + ACCEL)*3 + INTEG | 36 + (VOL*2 + ACCEL)*3 + INTEG.
namespace rt { extern const RenderKernelFn g_render_kernels_whitted[16], g_render_kernels_direct[16], g_render_kernels_path[16], g_render_kernels_weighted[8]; }
namespace rt { extern const PipeShadeFn g_pipe_shade_whitted[6], g_pipe_shade_direct[6], g_pipe_shade_path[6]; extern const PipeTraceFn g_pipe_trace[8]; extern const PipeShadeFn g_pipe_vertex[3];
               extern const PipeMarchFn g_pipe_march[6]; }
static RenderKernelFn render_kernel_of(int variant) {
    const RenderKernelFn *t = (variant % 3 == 0) ? g_render_kernels_whitted : (variant % 3 == 1) ? g_render_kernels_direct : g_render_kernels_path;
    return t[variant < 24 ? variant / 3 : variant < 36 ? 8 + (variant - 24) / 3 : 12 + (variant - 36) / 3];
}

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

// One record per primitive for the flat traversal (rt_device.h DevScene::ltris / lrefs / tnodes; the entry encoding: rt_traverse.h RT_LE_*).
// Rounds 2-5 kept one 48-byte copy per leaf REFERENCE, a leaf's copies side by side: 25.1 M copies of the benchmark soup's 1 M triangles (1.2 GB;
// 12 GB at 10 M triangles) that no cache level holds.  Now a primitive has ONE record, placed where the depth-first leaf walk first meets it (so the
// primitives of neighbouring leaves are neighbours), RT_TRI_STRIDE float4 units apart; a leaf node names its first primitive inline, a leaf of two
// the second one in its word 1, a larger leaf the index of its remaining entries in `lrefs` (the reference's own form, kdtree.cpp:55-64).
// `copies` (PBRT_HIP_LEAF_COPIES, measurements only): every reference gets a record of its own again -- the same kernel, the old footprint.
// `runs`: the leaves own runs of consecutive records and word 1 is the primitive count (DevScene::leaf_runs: rounds 2-5's layout, without the line
// alignment; what scenes of a few thousand references use -- cache resident, bound by instruction issue, where fetching entries costs 3 %).
struct LeafLayout {
    NodeVec tnodes;                       // the nodes with leaves in entry form
    RefVec lrefs;                         // entries of the third and later primitives of the leaves
    RefVec slot_prim;                     // record slot -> primitive
    size_t n_slots = 0;
};
static bool leaf_cursor_layout(const NodeVec &nodes, const RefVec &leaf_refs, uint32_t n_tris, bool copies, bool runs, LeafLayout &o) {
    copies = copies || runs;
    const size_t N = nodes.size();
    o.tnodes.resize(N);
    const size_t B = size_t(1) << 18, nb = (N + B - 1) / B;
    const size_t nthreads = nb < 4 ? 1 : std::min<size_t>(nb, std::max(1u, std::min(64u, std::thread::hardware_concurrency())));
    auto run = [&](auto fn) {
        if (nthreads == 1) { for (size_t b = 0; b < nb; ++b) fn(b); return; }
        std::atomic<size_t> next(0);
        ThreadGroup pool;
        for (size_t t = 0; t < nthreads; ++t) pool.spawn([&] { for (;;) { const size_t b = next.fetch_add(1); if (b >= nb) return; fn(b); } });
    };
    auto leaf_n = [&](const Node &n) -> uint32_t { return (n.x & 3u) == 3u ? n.x >> 2 : 0u; };
    auto ref = [&](const Node &n, uint32_t np, uint32_t k) -> uint32_t { return np == 1 ? n.y : leaf_refs[n.y + k]; };
    // pass 1: where the walk first meets every primitive (64-bit key = node << 32 | position in the leaf; minimum over its references)
    std::unique_ptr<std::atomic<uint64_t>[]> first;
    if (!copies) {
        first.reset(new std::atomic<uint64_t>[size_t(n_tris) + 1]);
        for (size_t i = 0; i <= n_tris; ++i) first[i].store(~0ull, std::memory_order_relaxed);
        run([&](size_t b) {
            const size_t hi = std::min(N, (b + 1) * B);
            for (size_t i = b * B; i < hi; ++i) {
                const Node n = nodes[i]; const uint32_t np = leaf_n(n);
                for (uint32_t k = 0; k < np; ++k) {
                    const uint64_t key = uint64_t(i) << 32 | k;
                    std::atomic<uint64_t> &f = first[ref(n, np, k)];
                    uint64_t cur = f.load(std::memory_order_relaxed);
                    while (key < cur && !f.compare_exchange_weak(cur, key, std::memory_order_relaxed)) {}
                }
            }
        });
    }
    // pass 2: per block of nodes, the records it opens and the list entries its leaves of three or more need
    std::vector<size_t> slots(nb + 1, 0), lists(nb + 1, 0);
    run([&](size_t b) {
        const size_t hi = std::min(N, (b + 1) * B);
        size_t ns = 0, nl = 0;
        for (size_t i = b * B; i < hi; ++i) {
            const Node n = nodes[i]; const uint32_t np = leaf_n(n);
            if (np >= 3 && !runs) nl += (np - 1 + 1) & ~size_t(1);            // lists start at even indices (the cursor is stored halved)
            if (copies) ns += np;
            else for (uint32_t k = 0; k < np; ++k) ns += first[ref(n, np, k)].load(std::memory_order_relaxed) == (uint64_t(i) << 32 | k);
        }
        slots[b + 1] = ns; lists[b + 1] = nl;
    });
    for (size_t b = 0; b < nb; ++b) { slots[b + 1] += slots[b]; lists[b + 1] += lists[b]; }
    o.n_slots = slots[nb];
    if (o.n_slots * RT_TRI_STRIDE >= RT_LE_POS || lists[nb] / 2 >= RT_LE_POS) return false;
    o.slot_prim.resize(o.n_slots);
    o.lrefs.resize(lists[nb] ? lists[nb] : 1);
    // pass 3: a primitive's slot (the record it shares, or one per reference)
    std::vector<uint32_t> slot_of;
    if (!copies) {
        slot_of.assign(size_t(n_tris) + 1, 0u);
        run([&](size_t b) {
            const size_t hi = std::min(N, (b + 1) * B);
            size_t at = slots[b];
            for (size_t i = b * B; i < hi; ++i) {
                const Node n = nodes[i]; const uint32_t np = leaf_n(n);
                for (uint32_t k = 0; k < np; ++k) {
                    const uint32_t p = ref(n, np, k);
                    if (first[p].load(std::memory_order_relaxed) == (uint64_t(i) << 32 | k)) { slot_of[p] = uint32_t(at); o.slot_prim[at++] = p; }
                }
            }
        });
    }
    // pass 4: the leaves in entry form
    run([&](size_t b) {
        const size_t hi = std::min(N, (b + 1) * B);
        size_t at = slots[b], lat = lists[b];
        for (size_t i = b * B; i < hi; ++i) {
            const Node n = nodes[i];
            o.tnodes[i] = n;
            if ((n.x & 3u) != 3u) continue;
            const uint32_t np = n.x >> 2;
            if (np == 0) { o.tnodes[i].x = RT_LE_NONE; o.tnodes[i].y = ~RT_LE_POS; continue; }       // entry RT_LE_NONE (runs: + a count that is never read)
            auto pos = [&](uint32_t k) -> uint32_t {
                if (copies) { o.slot_prim[at + k] = ref(n, np, k); return uint32_t(at + k) * RT_TRI_STRIDE; }
                return slot_of[ref(n, np, k)] * RT_TRI_STRIDE;
            };
            o.tnodes[i].x = pos(0) << 2 | 3u;
            if (runs) { for (uint32_t k = 1; k < np; ++k) pos(k); o.tnodes[i].y = (np > 1 ? RT_LE_MORE : 0u) | np; }
            else if (np == 1) o.tnodes[i].y = 0u;
            else if (np == 2) o.tnodes[i].y = RT_LE_MORE | pos(1);
            else {
                o.tnodes[i].y = RT_LE_MORE | RT_LE_LIST | uint32_t(lat / 2);
                for (uint32_t k = 1; k < np; ++k) o.lrefs[lat++] = pos(k) | (k + 1 < np ? RT_LE_MORE | RT_LE_LIST : 0u);
                if (lat & 1) o.lrefs[lat++] = RT_LE_NONE;                          // padding, never read
            }
            if (copies) at += np;
        }
    });
    if (lists[nb] == 0) o.lrefs[0] = 0u;
    return true;
}
// the records on the host (the check of rt::derive_leaf_records_kernel below: PBRT_HIP_VERIFY_DERIVED compares the two byte for byte)
static void leaf_records_fill_host(const RefVec &slot_prim, const std::vector<DevTri> &tris, std::vector<float4> &ltris) {
    ltris.assign(slot_prim.size() * RT_TRI_STRIDE + 4, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t i = 0; i < slot_prim.size(); ++i) {
        const uint32_t prim = slot_prim[i];
        float4 q2 = tris[prim].q2; std::memcpy(&q2.w, &prim, 4);
        float4 *dst = ltris.data() + i * RT_TRI_STRIDE;
        dst[0] = tris[prim].q0; dst[1] = tris[prim].q1; dst[2] = q2;
    }
}
// ... and on the device: one thread per record copies its primitive out of the mesh-order records that are in HBM anyway (the primitive's index goes
// into the spare word; the padding was zeroed by a memset before the launch)
namespace rt {
__global__ void derive_leaf_records_kernel(const unsigned *__restrict__ slot_prim, const DevTri *__restrict__ tris, float4 *__restrict__ ltris, size_t n_slots) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    const unsigned prim = slot_prim[i];
    const DevTri t = tris[prim];
    float4 q2 = t.q2; q2.w = __uint_as_float(prim);
    float4 *dst = ltris + i * RT_TRI_STRIDE;
    dst[0] = t.q0; dst[1] = t.q1; dst[2] = q2;
}
}  // namespace rt

// Triangle::Intersect's frame with the mesh's own uvs (trianglemesh.cpp:248-268 incl. the zero-determinant fallback through
// CoordinateSystem, geometry.h:324-334) + DifferentialGeometry ctor (shape.cpp:43-50): geometric normal, raw dpdu
static void host_tri_frame_uv(const float *v, const float *uv, bool flip, float nn[3], float dpdu[3]) {
    const float du1 = uv[0] - uv[4], du2 = uv[2] - uv[4], dv1 = uv[1] - uv[5], dv2 = uv[3] - uv[5];
    const float determinant = du1 * dv2 - dv1 * du2;
    float dpdv[3];
    if (determinant == 0.f) {
        const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
        float c[3] = {(e2[1] * e1[2]) - (e2[2] * e1[1]), (e2[2] * e1[0]) - (e2[0] * e1[2]), (e2[0] * e1[1]) - (e2[1] * e1[0])};
        const float inv = 1.f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        const float v1[3] = {c[0] * inv, c[1] * inv, c[2] * inv};
        if (fabsf(v1[0]) > fabsf(v1[1])) { const float invLen = 1.f / sqrtf(v1[0] * v1[0] + v1[2] * v1[2]); dpdu[0] = -v1[2] * invLen; dpdu[1] = 0.f; dpdu[2] = v1[0] * invLen; }
        else { const float invLen = 1.f / sqrtf(v1[1] * v1[1] + v1[2] * v1[2]); dpdu[0] = 0.f; dpdu[1] = v1[2] * invLen; dpdu[2] = -v1[1] * invLen; }
        dpdv[0] = (v1[1] * dpdu[2]) - (v1[2] * dpdu[1]); dpdv[1] = (v1[2] * dpdu[0]) - (v1[0] * dpdu[2]); dpdv[2] = (v1[0] * dpdu[1]) - (v1[1] * dpdu[0]);
    } else {
        const float invdet = 1.f / determinant;
        for (int a = 0; a < 3; ++a) {
            const float dp1 = v[a] - v[6 + a], dp2 = v[3 + a] - v[6 + a];
            dpdu[a] = ((dv2 * dp1) - (dv1 * dp2)) * invdet;
            dpdv[a] = ((-du2 * dp1) + (du1 * dp2)) * invdet;
        }
    }
    const float c[3] = {(dpdu[1] * dpdv[2]) - (dpdu[2] * dpdv[1]), (dpdu[2] * dpdv[0]) - (dpdu[0] * dpdv[2]), (dpdu[0] * dpdv[1]) - (dpdu[1] * dpdv[0])};
    const float inv = 1.f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (int a = 0; a < 3; ++a) { nn[a] = c[a] * inv; if (flip) nn[a] = -1.f * nn[a]; }
}

// The tree as sibling pairs (rt_device.h DevScene::tpairs: a record holds the two node words of a node's below child and the two of its above
// child, addressed by absolute index) laid out in BLOCKS for the two-level step (kdp_step, rt_traverse.h): an "owner" node P is followed by the pairs of
// its interior children -- {pair(P), pair(below(P)), pair(above(P))}, 16 / 32 / 48 bytes, never across a 64-byte boundary (next-fit
// padding) -- and bits 30 / 31 of every word 1 that points at P say which of the two follow.  The owners are the root and, recursively,
// the interior grandchildren of an owner; the nodes in between are "members" of their parent's block (flags 0: when a member is reached
// through a pop it takes a one-level step).  Blocks are emitted depth-first, the below side first, so a subtree stays contiguous.
// Two phases: pair_blocks_order() decides where every pair goes (a sequential depth-first walk that looks at the tree's SHAPE only, so it runs
// beside leaf_cursor_layout on another thread), pair_blocks_fill() writes the records (needs the leaves in entry form; 64 threads).
// Round 5: the blocks of the tree's TOP levels come first, breadth-first (owner level by owner level, below side first) and packed without
// padding, RT_TOP_PREFIX records at most (any prefix of the array is "the topmost blocks": what an LDS copy would want -- measured, not kept,
// profiles/r05_lds_top_scan.txt -- and what every ray walks sits in 64 KB).  The subtrees below that frontier follow depth-first in 64-byte-aligned blocks as before.
#ifndef RT_TOP_PREFIX
#define RT_TOP_PREFIX 4095u          // 1365 blocks of three records: 11-12 levels of a full tree
#endif
struct PairBlockOrder { std::vector<uint32_t> order, pos; std::vector<uint8_t> owner; uint32_t top = 0; };
static void pair_blocks_order(const NodeVec &tn, PairBlockOrder &o) {
    o.order.clear(); o.pos.clear(); o.owner.clear(); o.top = 0;
    if (tn.empty() || (tn[0].x & 3u) == 3u) return;
    auto interior = [&](uint32_t n) { return (tn[n].x & 3u) != 3u; };
    std::vector<uint32_t> &order = o.order;                                   // parent node of each emitted pair (~0u = padding)
    o.pos.assign(tn.size(), ~0u);                                             // node -> index of its children's pair
    o.owner.assign(tn.size(), 0);
    // emit the block of owner P behind `ord`; `next` receives the owners below it (the interior children of its members), below side first
    auto emit = [&](std::vector<uint32_t> &ord, uint32_t P, bool aligned, std::vector<uint32_t> &next, bool reversed) {
        const uint32_t b = P + 1u, a = tn[P].y;
        const bool bI = interior(b), aI = interior(a);
        const size_t size = 1u + (bI ? 1u : 0u) + (aI ? 1u : 0u);
        if (aligned && (ord.size() % 4) + size > 4) while (ord.size() % 4) ord.push_back(~0u);
        o.owner[P] = 1;
        ord.push_back(P);
        if (bI) ord.push_back(b);
        if (aI) ord.push_back(a);
        const uint32_t mem[2] = {reversed ? a : b, reversed ? b : a};
        const bool memI[2] = {reversed ? aI : bI, reversed ? bI : aI};
        for (int k = 0; k < 2; ++k) {
            if (!memI[k]) continue;
            const uint32_t m = mem[k], mb = m + 1u, ma = tn[m].y;
            const uint32_t c[2] = {reversed ? ma : mb, reversed ? mb : ma};
            for (int j = 0; j < 2; ++j) if (interior(c[j])) next.push_back(c[j]);
        }
    };
    // the top: breadth-first, dense
    std::vector<uint32_t> level{0u}, below;
    size_t li = 0;
    while (li < level.size() && order.size() + 3u <= RT_TOP_PREFIX) {
        emit(order, level[li++], false, below, false);
        if (li == level.size()) { level.swap(below); below.clear(); li = 0; }
    }
    o.top = uint32_t(order.size());
    while (order.size() % 4) order.push_back(~0u);
    for (size_t i = 0; i < order.size(); ++i) if (order[i] != ~0u) o.pos[order[i]] = uint32_t(i);
    // what is left: the rest of the current level, then the owners found below it.  Each of these frontier subtrees is laid out depth-first on its own (a stack:
    // below(below(P)) follows P), starting on a 64-byte boundary -- so its layout depends on nothing outside it, and the subtrees are walked by all threads (the walk
    // over 123 M interior nodes took 2 s at 10 M triangles); the pieces follow the prefix in frontier order whatever the thread count.
    std::vector<uint32_t> roots;
    for (size_t k = li; k < level.size(); ++k) roots.push_back(level[k]);
    for (uint32_t r : below) roots.push_back(r);
    std::vector<std::vector<uint32_t>> sub(roots.size());
    const size_t nthreads = tn.size() < (size_t(1) << 22) ? 1 : std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    auto run = [&](auto fn) {
        if (nthreads == 1) { for (size_t k = 0; k < roots.size(); ++k) fn(k); return; }
        std::atomic<size_t> next(0);
        ThreadGroup pool;
        for (size_t t = 0; t < nthreads; ++t) pool.spawn([&] { for (;;) { const size_t k = next.fetch_add(1); if (k >= roots.size()) return; fn(k); } });
    };
    run([&](size_t k) {
        std::vector<uint32_t> &ord = sub[k], todo{roots[k]};
        while (!todo.empty()) { const uint32_t P = todo.back(); todo.pop_back(); emit(ord, P, true, todo, true); }
        while (ord.size() % 4) ord.push_back(~0u);
    });
    std::vector<size_t> base(roots.size() + 1, order.size());
    for (size_t k = 0; k < roots.size(); ++k) base[k + 1] = base[k] + sub[k].size();
    order.resize(base[roots.size()]);
    run([&](size_t k) {
        const std::vector<uint32_t> &ord = sub[k];
        uint32_t *dst = order.data() + base[k];
        for (size_t i = 0; i < ord.size(); ++i) { dst[i] = ord[i]; if (ord[i] != ~0u) o.pos[ord[i]] = uint32_t(base[k] + i); }
        std::vector<uint32_t>().swap(sub[k]);
    });
}
// `tn`: the nodes with the leaves in entry form (LeafLayout::tnodes; interior nodes as in the tree: same shape as pair_blocks_order saw)
static void pair_blocks_fill(const NodeVec &tn, const PairBlockOrder &o, std::vector<uint4> &pairs, uint32_t &root_x, uint32_t &root_y) {
    pairs.clear();
    if (tn.empty()) { root_x = 3u; root_y = 0u; pairs.push_back(make_uint4(3u, 0u, 3u, 0u)); return; }
    root_x = tn[0].x;
    if ((tn[0].x & 3u) == 3u) { root_y = tn[0].y; pairs.push_back(make_uint4(3u, 0u, 3u, 0u)); return; }
    const std::vector<uint32_t> &order = o.order;
    if (order.size() >= (size_t(1) << 30)) return;
    auto interior = [&](uint32_t n) { return (tn[n].x & 3u) != 3u; };
    auto word1 = [&](uint32_t n) -> uint32_t {
        if (!interior(n)) return tn[n].y;                                     // leaf: flags of its first entry | cursor
        uint32_t y = o.pos[n];
        if (o.owner[n]) y |= (interior(n + 1u) ? 1u << 30 : 0u) | (interior(tn[n].y) ? 1u << 31 : 0u);
        return y;
    };
    pairs.resize(order.size());
    auto fill = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint32_t P = order[i];
            if (P == ~0u) { pairs[i] = make_uint4(3u, 0u, 3u, 0u); continue; }
            const uint32_t b = P + 1u, a = tn[P].y;
            pairs[i] = make_uint4(tn[b].x, word1(b), tn[a].x, word1(a));
        }
    };
    const size_t nthreads = order.size() < (size_t(1) << 20) ? 1 : std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    if (nthreads == 1) fill(0, order.size());
    else {
        ThreadGroup pool;
        for (size_t t = 0; t < nthreads; ++t) pool.spawn(fill, order.size() * t / nthreads, order.size() * (t + 1) / nthreads);
    }
    root_y = word1(0u);
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

// (re)allocate a scratch buffer that is only ever used inside one rt_render call
template <class T>
static int ensure(RtScene *s, T **buf, size_t *cap, size_t need) {
    if (need <= *cap) return RT_OK;
    if (*buf) { HIPCHK(hipStreamSynchronize(s->stream)); HIPWARN(hipFree(*buf)); *buf = nullptr; *cap = 0; }
    HIPCHK(hipMalloc((void **)buf, need * sizeof(T)));
    *cap = need;
    return RT_OK;
}

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

static void fill_info(const KdTree &tree, const GridAccelData &g, int kind, uint32_t n_tris, RtAccelInfo *info);
extern "C" int rt_scene_destroy(RtScene *s);

// No C++ exception crosses the C boundary: the host builders allocate gigabytes and start worker threads (every group of them is joined while the
// exception unwinds, rt_internal.h ThreadGroup), so bad_alloc / a failed thread start end in a status code, not in std::terminate.
template <class F> static int guarded(const char *what, F &&f) {
    try { return f(); }
    catch (const std::bad_alloc &) { return fail(RT_ENOMEM, std::string(what) + ": out of host memory"); }
    catch (const std::exception &e) { return fail(RT_ESTATE, std::string(what) + ": " + e.what()); }
}

extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(int *count) {
    if (!count) return fail(RT_EINVAL, "rt_device_count: null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RT_EDEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n; return RT_OK;
}

// structural check of an accelerator handed in by the caller (rt_scene_create_prebuilt): every index the traversal follows stays in range
static int check_prebuilt(const RtPrebuiltAccel *a, uint32_t n_tris) {
    if (!a->nodes || (a->n_leaf_refs && !a->leaf_refs)) return fail(RT_EINVAL, "rt_scene_create_prebuilt: null accelerator arrays");
    const uint32_t *nd = a->nodes;
    for (uint32_t i = 0; i < a->n_leaf_refs; ++i) if (a->leaf_refs[i] >= n_tris) return fail(RT_EINVAL, "rt_scene_create_prebuilt: primitive index out of range");
    if (a->kind == RT_ACCEL_KDTREE) {
        if (a->n_nodes == 0 && n_tris != 0) return fail(RT_EINVAL, "rt_scene_create_prebuilt: empty tree");
        // The per-thread spill area of the traversal stack is sized from max_depth (scene_create), so the claim is checked, not trusted: a
        // child's index is larger than its parent's, hence one forward sweep gives every node's depth (the deeper path wins if a node has two parents).
        if (a->max_depth > 64) return fail(RT_EINVAL, "rt_scene_create_prebuilt: max_depth beyond 64");
        std::vector<uint8_t> depth(a->n_nodes, 0);
        for (uint32_t i = 0; i < a->n_nodes; ++i) {
            const uint32_t x = nd[2 * size_t(i)], y = nd[2 * size_t(i) + 1];
            if ((x & 3u) != 3u) {
                if (y <= i + 1u || y >= a->n_nodes || i + 1u >= a->n_nodes) return fail(RT_EINVAL, "rt_scene_create_prebuilt: child index out of range");
                if (!std::isfinite(*reinterpret_cast<const float *>(&nd[2 * size_t(i)]))) return fail(RT_EINVAL, "rt_scene_create_prebuilt: split position is not finite");
                const unsigned dc = unsigned(depth[i]) + 1u;
                if (dc > a->max_depth) return fail(RT_EINVAL, "rt_scene_create_prebuilt: the tree is deeper than its max_depth says");
                if (depth[i + 1u] < dc) depth[i + 1u] = uint8_t(dc);
                if (depth[y] < dc) depth[y] = uint8_t(dc);
            } else {
                const uint32_t np = x >> 2;
                if (np == 1u ? y >= n_tris : (np > 1u && (y > a->n_leaf_refs || np > a->n_leaf_refs - y))) return fail(RT_EINVAL, "rt_scene_create_prebuilt: leaf list out of range");
            }
        }
    } else {
        const unsigned long long nv = (unsigned long long)a->grid_nvoxels[0] * a->grid_nvoxels[1] * a->grid_nvoxels[2];
        if (a->grid_nvoxels[0] < 1 || a->grid_nvoxels[1] < 1 || a->grid_nvoxels[2] < 1 || nv != a->n_nodes) return fail(RT_EINVAL, "rt_scene_create_prebuilt: voxel counts do not match");
        for (uint32_t i = 0; i < a->n_nodes; ++i) {
            const uint32_t off = nd[2 * size_t(i)], cnt = nd[2 * size_t(i) + 1];
            if (off > a->n_leaf_refs || cnt > a->n_leaf_refs - off) return fail(RT_EINVAL, "rt_scene_create_prebuilt: voxel list out of range");
        }
    }
    for (int k = 0; k < 6; ++k) if (!std::isfinite(a->bounds[k])) return fail(RT_EINVAL, "rt_scene_create_prebuilt: bounds are not finite");
    for (int k = 0; k < 3; ++k) {
        if (a->bounds[k] > a->bounds[3 + k] && n_tris != 0) return fail(RT_EINVAL, "rt_scene_create_prebuilt: bounds are inverted");
        // (a flat scene has width = inv_width = 0 on its thin axis, as GridAccel's constructor makes them: grid.cpp:102-104)
        if (a->kind == RT_ACCEL_GRID && (!(a->grid_width[k] >= 0.f) || !(a->grid_inv_width[k] >= 0.f) || !std::isfinite(a->grid_width[k]) || !std::isfinite(a->grid_inv_width[k])))
            return fail(RT_EINVAL, "rt_scene_create_prebuilt: voxel widths must be non-negative and finite");
    }
    return RT_OK;
}

static int scene_create(const RtSceneDesc *d, int device, const RtPrebuiltAccel *pre, RtScene **out);
int rt_scene_create(const RtSceneDesc *d, int device, RtScene **out) { return guarded("rt_scene_create", [&] { return scene_create(d, device, nullptr, out); }); }
// The same scene with the accelerator somebody else built (rt_accel_build / rt_scene_accel_copy of another rank's scene): the ranks of one
// node build the kd-tree ONCE (10 M triangles: 15 s on all host cores) instead of once per process.  The arrays are the canonical flattened
// tree (pbrt_hip.h RtAccelInfo / rt_accel_copy); everything the device derives from them (leaf-ordered records, pair blocks) is rebuilt here.
int rt_scene_create_prebuilt(const RtSceneDesc *d, int device, const RtPrebuiltAccel *pre, RtScene **out) {
    if (!pre) return fail(RT_EINVAL, "rt_scene_create_prebuilt: null accelerator");
    return guarded("rt_scene_create_prebuilt", [&] { return scene_create(d, device, pre, out); });
}
static int scene_create(const RtSceneDesc *d, int device, const RtPrebuiltAccel *pre, RtScene **out) {
    if (!d || !out) return fail(RT_EINVAL, "rt_scene_create: null argument");
    if (pre) {
        if (pre->kind != d->accel.kind) return fail(RT_EINVAL, "rt_scene_create_prebuilt: accelerator kind differs from the scene's");
        int rc = check_prebuilt(pre, d->n_tris); if (rc) return rc;
    }
    if (d->n_tris && (!d->tri_verts || !d->tri_material || !d->tri_light || !d->tri_flags))
        return fail(RT_EINVAL, "rt_scene_create: missing triangle arrays");
    if (d->accel.kind != RT_ACCEL_KDTREE && d->accel.kind != RT_ACCEL_GRID) return fail(RT_EINVAL, "rt_scene_create: unknown accelerator kind");
    for (uint32_t i = 0; i < d->n_tris; ++i) {
        if (d->tri_material[i] >= d->n_materials) return fail(RT_EINVAL, "rt_scene_create: material index out of range");
        const int32_t tl = d->tri_light[i];               // the device indexes `lights` with it (make_vertex, prim_normal_light)
        if (tl < -1 || tl >= int32_t(d->n_lights) || (tl >= 0 && d->lights[tl].type != RT_LIGHT_AREA))
            return fail(RT_EINVAL, "rt_scene_create: triangle refers to a light that is out of range or not an area light");
    }
    if (d->n_lights > 65534u) return fail(RT_EINVAL, "rt_scene_create: more than 65534 lights");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(RT_EDEVICE, "rt_scene_create: no HIP device visible (the product path has no CPU fallback)");
    // every error exit below goes through the guard: rt_scene_destroy frees whatever has been created so far
    struct Guard { RtScene *p; ~Guard() { if (p) rt_scene_destroy(p); } } guard{new RtScene()};
    RtScene *s = guard.p;
    if (device >= 0) { hipError_t e = hipSetDevice(device); if (e != hipSuccess) return fail(RT_EDEVICE, "hipSetDevice failed"); }
    HIPCHK(hipGetDevice(&s->device));
    HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)); s->own_stream = true;
    HIPCHK(hipEventCreate(&s->ev0)); HIPCHK(hipEventCreate(&s->ev1));
    s->n_tris = d->n_tris;

    const bool tlog = knob("PBRT_HIP_CREATE_LOG") != nullptr;           // where a scene create spends its time (10 M triangles: a minute)
    auto t_prev = std::chrono::steady_clock::now();
    auto tick = [&](const char *what) {
        if (!tlog) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "CREATE %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count()); t_prev = now;
    };
    s->accel_kind = d->accel.kind;
    if (pre) {
        const Node *pn = reinterpret_cast<const Node *>(pre->nodes);
        s->tree.nodes.assign(pn, pn + pre->n_nodes); s->tree.leaf_refs.assign(pre->leaf_refs, pre->leaf_refs + pre->n_leaf_refs);
        s->tree.max_depth = int(pre->max_depth); s->tree.build_seconds = 0.0;
        std::memcpy(s->tree.bounds, pre->bounds, sizeof s->tree.bounds);
        if (s->accel_kind == RT_ACCEL_GRID) {
            s->gridacc.voxels = s->tree.nodes; s->gridacc.refs = s->tree.leaf_refs; s->gridacc.build_seconds = 0.0;
            std::memcpy(s->gridacc.bounds, pre->bounds, sizeof s->gridacc.bounds);
            for (int a = 0; a < 3; ++a) { s->gridacc.nvox[a] = pre->grid_nvoxels[a]; s->gridacc.width[a] = pre->grid_width[a]; s->gridacc.inv_width[a] = pre->grid_inv_width[a]; }
        }
    } else if (s->accel_kind == RT_ACCEL_GRID) {
        build_grid(d->tri_verts, d->n_tris, s->gridacc);
        s->tree.nodes = s->gridacc.voxels; s->tree.leaf_refs = s->gridacc.refs; s->tree.max_depth = 0;
        std::memcpy(s->tree.bounds, s->gridacc.bounds, sizeof s->tree.bounds); s->tree.build_seconds = s->gridacc.build_seconds;
    } else build_kdtree(d->tri_verts, d->n_tris, d->accel, s->tree);

    tick("accelerator");
    // the order of the pair blocks: a sequential walk over the tree's shape (1.8 s at 10 M triangles) on its own thread, beside everything up to the pair fill
    PairBlockOrder pbo;
    std::thread order_thread;
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{order_thread};
    if (s->accel_kind == RT_ACCEL_KDTREE) order_thread = std::thread([&] { pair_blocks_order(s->tree.nodes, pbo); });
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
    std::vector<DevTriShading> dshading; std::vector<int> shading_idx;
    if (d->tri_shading) {
        if (d->n_shading && !d->shading) return fail(RT_EINVAL, "rt_scene_create: tri_shading without shading records");
        shading_idx.assign(d->n_tris, -1);
    }
    for (uint32_t i = 0; i < d->n_tris; ++i) {
        float nn[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f};
        const int sh = (d->tri_shading && !(d->tri_flags[i] & 2u)) ? d->tri_shading[i] : -1;
        bool smooth = false;
        if (sh >= 0) {                                   // the mesh has uv / N / S: the frame depends on its uvs (trianglemesh.cpp:248-268)
            if (uint32_t(sh) >= d->n_shading) return fail(RT_EINVAL, "rt_scene_create: shading record index out of range");
            const RtTriShading &r = d->shading[sh];
            if ((r.flags & (RT_SHADING_N | RT_SHADING_S)) && (r.xform >= d->n_xforms || !d->xforms)) return fail(RT_EINVAL, "rt_scene_create: shading record refers to a transform out of range");
            float dpdu[3];
            host_tri_frame_uv(d->tri_verts + size_t(9) * i, r.uv, (d->tri_flags[i] & 1u) != 0, nn, dpdu);
            const float inv = 1.f / sqrtf(dpdu[0] * dpdu[0] + dpdu[1] * dpdu[1] + dpdu[2] * dpdu[2]);
            for (int a = 0; a < 3; ++a) sn[a] = dpdu[a] * inv;
            if (r.flags & (RT_SHADING_N | RT_SHADING_S)) {
                smooth = true; s->has_ext = true;
                DevTriShading o; std::memset(&o, 0, sizeof o);
                o.flags = r.flags; o.xform = r.xform;
                std::memcpy(o.uv, r.uv, sizeof o.uv); std::memcpy(o.dpdu, dpdu, sizeof o.dpdu);
                std::memcpy(o.n, r.n, sizeof o.n); std::memcpy(o.s, r.s, sizeof o.s);
                shading_idx[i] = int(dshading.size()); dshading.push_back(o);
            }
        } else if (!(d->tri_flags[i] & 2u)) host_tri_frame(d->tri_verts + size_t(9) * i, (d->tri_flags[i] & 1u) != 0, nn, sn);
        uint32_t bits = uint32_t(d->tri_material[i]) | (uint32_t(d->tri_flags[i] & 1u) << 16) | ((d->tri_flags[i] & 2u) ? RT_PRIM_QUADRIC : 0u) |
                        (smooth ? RT_PRIM_SHADING : 0u);
        int32_t light = d->tri_light[i];
        float fb, fl; std::memcpy(&fb, &bits, 4); std::memcpy(&fl, &light, 4);
        shade[2 * i] = make_float4(nn[0], nn[1], nn[2], fb);
        shade[2 * i + 1] = make_float4(sn[0], sn[1], sn[2], fl);
    }
    int rc;
    if ((rc = upload(s, shade.data(), shade.size(), &s->dev.tri_shade))) return rc;
    if (!dshading.empty()) {
        if ((rc = upload(s, shading_idx.data(), shading_idx.size(), &s->dev.tri_shading_idx))) return rc;
        if ((rc = upload(s, dshading.data(), dshading.size(), &s->dev.tri_shading))) return rc;
        if ((rc = upload(s, d->xforms, size_t(d->n_xforms) * 32, &s->dev.xforms))) return rc;
    }
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
    tick("triangle / shading records");
    // nodes (+ one node of padding: the traversal may fetch node i+1 together with node i) and the leaf lists
    auto upload_nodes = [&](const NodeVec &v, const uint2 **dev) -> int {
        void *p = nullptr;
        HIPCHK(hipMalloc(&p, (v.size() + 1) * sizeof(uint2)));
        s->allocs.push_back(p);
        if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(uint2), hipMemcpyHostToDevice));
        const uint2 pad = make_uint2(3u, 0u);
        HIPCHK(hipMemcpy((uint2 *)p + v.size(), &pad, sizeof pad, hipMemcpyHostToDevice));
        *dev = (const uint2 *)p;
        return RT_OK;
    };
    const uint2 *nodes_dev = nullptr;
    if ((rc = upload_nodes(s->tree.nodes, &nodes_dev))) return rc;
    if ((rc = upload(s, s->tree.leaf_refs.data(), s->tree.leaf_refs.size(), &s->dev.leaf_refs))) return rc;
    s->dev.nodes = nodes_dev;
    s->dev.tnodes = nodes_dev;
    if (s->accel_kind == RT_ACCEL_KDTREE) {
        LeafLayout ll;
        tick("node / leaf-list upload");
        // runs of consecutive records per leaf for scenes of a few thousand references (C2's 14 triangles: cache resident, bound by instruction issue -- the entry
        // form costs it 2.8 %, profiles/r06_dedup_scan.txt); everything larger shares one record per primitive
        bool runs = s->tree.leaf_refs.size() + s->tree.nodes.size() / 2 <= 32768;
        if (const char *e = knob("PBRT_HIP_LEAF_RUNS")) runs = std::atoi(e) != 0;
        if (!leaf_cursor_layout(s->tree.nodes, s->tree.leaf_refs, d->n_tris, knob("PBRT_HIP_LEAF_COPIES") != nullptr, runs, ll))
            return fail(RT_EINVAL, "rt_scene_create: primitive records beyond 2^30 float4 units or leaf entries beyond 2^31");
        s->dev.leaf_runs = runs ? 1u : 0u;
        const NodeVec &tn = ll.tnodes;
        tick("leaf entries");
        if ((rc = upload_nodes(tn, &s->dev.tnodes))) return rc;
        if ((rc = upload(s, ll.lrefs.data(), ll.lrefs.size(), &s->dev.lrefs))) return rc;
        {
            const unsigned *slot_prim_dev = nullptr;
            if ((rc = upload(s, ll.slot_prim.data(), ll.slot_prim.size(), &slot_prim_dev))) return rc;
            const size_t units = ll.n_slots * RT_TRI_STRIDE + 4;            // (+ one record of padding: a lane without a primitive never loads, but the array is never empty)
            void *p = nullptr;
            HIPCHK(hipMalloc(&p, units * sizeof(float4)));
            s->allocs.push_back(p);
            s->dev.ltris = (const float4 *)p;
            HIPCHK(hipMemsetAsync(p, 0, units * sizeof(float4), s->stream));
            if (ll.n_slots) hipLaunchKernelGGL(derive_leaf_records_kernel, dim3(unsigned((ll.n_slots + 255) / 256)), dim3(256), 0, s->stream, slot_prim_dev,
                                               (const DevTri *)s->dev.tris, (float4 *)p, ll.n_slots);
            HIPCHK(hipGetLastError());
            if (knob("PBRT_HIP_VERIFY_DERIVED")) {             // tests: the device fill against the host fill, byte for byte
                std::vector<float4> lt, back(units); leaf_records_fill_host(ll.slot_prim, tris, lt);
                HIPCHK(hipStreamSynchronize(s->stream));
                HIPCHK(hipMemcpy(back.data(), p, units * sizeof(float4), hipMemcpyDeviceToHost));
                if (lt.size() != units || std::memcmp(back.data(), lt.data(), units * sizeof(float4)) != 0) return fail(RT_ESTATE, "rt_scene_create: the device-built primitive records differ from the host fill");
            }
            s->n_leaf_tri_units = units;
            s->n_leaf_entries = ll.lrefs.size();
        }
        tick("primitive records (device)");
        std::vector<uint4> pairs;
        order_thread.join();
        pair_blocks_fill(tn, pbo, pairs, s->dev.root_x, s->dev.root_y);
        s->dev.top_pairs = pbo.top;
        if (pairs.empty() || pairs.size() >= (size_t(1) << 30)) return fail(RT_EINVAL, "rt_scene_create: pair records beyond 2^30");
        tick("pair blocks");
        if ((rc = upload(s, pairs.data(), pairs.size(), &s->dev.tpairs))) return rc;
        tick("pair upload");
    }
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
    {
        std::vector<unsigned> flags(d->n_lights ? d->n_lights : 1, 0u);
        s->n_drawing_lights = 0;
        for (uint32_t i = 0; i < d->n_lights; ++i) {       // ShapeSet::Sample (shape.h:115-121) draws one RandomFloat() when the emitter has several triangles
            const int draws = (d->lights[i].type == RT_LIGHT_AREA && d->lights[i].quadric_plus1 == 0 && d->lights[i].n_tris > 1) ? 1 : 0;
            if (i == 0) s->light_draws = draws; else if (draws != s->light_draws) s->light_draws = -1;
            flags[i] = unsigned(draws); s->n_drawing_lights += unsigned(draws);
        }
        if ((rc = upload(s, flags.data(), flags.size(), &s->light_draw_flags))) return rc;      // (read by the recurrence of a "weighted" frame with lights of mixed RNG use)
    }

    s->dev.n_tris = d->n_tris; s->dev.n_lights = d->n_lights;
    s->dev.accel_kind = s->accel_kind;
    for (int a = 0; a < 3; ++a) { s->dev.nvox[a] = s->gridacc.nvox[a]; s->dev.gwidth[a] = s->gridacc.width[a]; s->dev.ginv_width[a] = s->gridacc.inv_width[a]; }
    std::memcpy(s->dev.bounds, s->tree.bounds, sizeof s->dev.bounds);
    s->dev.cam = d->camera; s->dev.vol = d->volume; s->volume = d->volume;

    // persistent launch geometry: as many resident blocks as the kernel's registers/LDS admit
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, s->device));
    s->n_cus = prop.multiProcessorCount;
    {
        unsigned mx = 0;
        for (int k = 0; k < 48; ++k) {
            int per_cu = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)render_kernel_of(k), RT_BLOCK, 0));
            if (per_cu < 1) per_cu = 1;
            s->grids[k] = unsigned(prop.multiProcessorCount) * unsigned(per_cu);
            mx = s->grids[k] > mx ? s->grids[k] : mx;
        }
        for (int k = 0; k < 8; ++k) {
            int per_cu = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)g_render_kernels_weighted[k], RT_BLOCK, 0));
            s->wgrids[k] = unsigned(prop.multiProcessorCount) * unsigned(per_cu < 1 ? 1 : per_cu);
            mx = s->wgrids[k] > mx ? s->wgrids[k] : mx;
        }
        s->grid = mx;
    }
    s->n_threads = s->grid * RT_BLOCK;
    s->spill_depth = s->tree.max_depth > RT_TRACE_STACK ? s->tree.max_depth - RT_TRACE_STACK + 1 : 1;     // RT_TRACE_STACK <= RT_STACK_LDS
    HIPCHK(hipMalloc((void **)&s->work_counter, 64 * sizeof(unsigned long long)));      // 8 band counters, one 64-byte line each (the pipeline uses the first)
    HIPCHK(hipMalloc((void **)&s->counters, 64 * sizeof(unsigned long long)));        // 8 RtCounters, 16 RT_PROFILE, 2 x 16 RT_PROFILE_STAGES
    HIPCHK(hipMemsetAsync(s->counters, 0, 64 * sizeof(unsigned long long), s->stream));
    HIPCHK(hipMalloc((void **)&s->filter_dev, 256 * sizeof(float)));
    HIPCHK(hipMalloc((void **)&s->dev_scene, sizeof(DevScene)));
    HIPCHK(hipMalloc((void **)&s->dev_frame, sizeof(DevFrame)));
    HIPCHK(hipMemcpy(s->dev_scene, &s->dev, sizeof(DevScene), hipMemcpyHostToDevice));
    HIPCHK(hipEventCreate(&s->ev2));
    for (int k = 0; k < 8; ++k) {
        int per_cu = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)g_pipe_trace[k], RT_BLOCK, 0));
        if (const char *e = knob("PBRT_HIP_TRACE_BLOCKS_PER_CU")) per_cu = std::min(per_cu, std::max(1, std::atoi(e)));   // occupancy experiments
        s->trace_grids[k] = unsigned(prop.multiProcessorCount) * unsigned(per_cu < 1 ? 1 : per_cu);
        if (s->trace_grids[k] * RT_BLOCK > s->n_threads) s->n_threads = s->trace_grids[k] * RT_BLOCK;      // the spill area is shared
    }
    for (int k = 0; k < 6; ++k) {
        int per_cu = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)g_pipe_march[k], RT_BLOCK, 0));
        s->march_grids[k] = unsigned(prop.multiProcessorCount) * unsigned(per_cu < 1 ? 1 : per_cu);
        if (s->march_grids[k] * RT_BLOCK > s->n_threads) s->n_threads = s->march_grids[k] * RT_BLOCK;
    }
    HIPCHK(hipMalloc((void **)&s->spill, size_t(s->spill_depth) * s->n_threads * sizeof(uint4)));     // uint4 entries in the pair form, uint2 otherwise
    HIPCHK(hipMalloc((void **)&s->dev_pool, sizeof(PipePool)));
    HIPCHK(hipMalloc((void **)&s->trace_qc, RT_QC_STRIDE * sizeof(unsigned)));
    HIPCHK(hipHostMalloc((void **)&s->h_qcount, size_t(RT_PIPE_QN) * RT_QC_STRIDE * sizeof(unsigned), hipHostMallocDefault));
    HIPCHK(hipStreamSynchronize(s->stream));
    guard.p = nullptr;
    *out = s;
    return RT_OK;
}

int rt_scene_destroy(RtScene *s) {
    if (!s) return RT_OK;
    HIPWARN(hipSetDevice(s->device));
    if (s->stream) hipStreamSynchronize(s->stream);
    for (void *p : s->allocs) HIPWARN(hipFree(p));
    if (s->own_accum && s->accum) HIPWARN(hipFree(s->accum));
    HIPWARN(hipFree(s->spill)); HIPWARN(hipFree(s->work_counter)); HIPWARN(hipFree(s->counters)); HIPWARN(hipFree(s->filter_dev));
    if (s->frames) HIPWARN(hipFree(s->frames));
    if (s->samples) HIPWARN(hipFree(s->samples));
    if (s->resolve_buf) HIPWARN(hipFree(s->resolve_buf));
    if (s->vol_buf) HIPWARN(hipFree(s->vol_buf));
    if (s->light_dims) HIPWARN(hipFree(s->light_dims));
    if (s->wt_base) HIPWARN(hipFree(s->wt_base));
    if (s->wt_recbase) HIPWARN(hipFree(s->wt_recbase));
    if (s->wt_rec) HIPWARN(hipFree(s->wt_rec));
    if (s->wt_pick) HIPWARN(hipFree(s->wt_pick));
    if (s->wt_total) HIPWARN(hipHostFree(s->wt_total));
    if (s->wt_sums) HIPWARN(hipFree(s->wt_sums));
    for (hipEvent_t e : s->wt_ev) if (e) HIPWARN(hipEventDestroy(e));
    HIPWARN(hipFree(s->dev_scene)); HIPWARN(hipFree(s->dev_frame));
    HIPWARN(hipFree(s->pool.state)); HIPWARN(hipFree(s->pool.ray_o)); HIPWARN(hipFree(s->pool.hit)); HIPWARN(hipFree(s->pool.q_o));
    HIPWARN(hipFree(s->pool.q_slot)); HIPWARN(hipFree(s->pool.q_count)); HIPWARN(hipFree(s->pool.wave_work)); HIPWARN(hipFree(s->dev_pool));
    HIPWARN(hipFree(s->trace_buf)); HIPWARN(hipFree(s->trace_qc));
    if (s->h_qcount) HIPWARN(hipHostFree(s->h_qcount));
    for (hipEvent_t e : s->pipe_ev) HIPWARN(hipEventDestroy(e));
    for (hipEvent_t e : s->pipe_fence) HIPWARN(hipEventDestroy(e));
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
    if (p.kind != RT_ACCEL_GRID && p.kind != RT_ACCEL_KDTREE) return fail(RT_EINVAL, "rt_accel_build: unknown accelerator kind");
    return guarded("rt_accel_build", [&] {
        std::unique_ptr<RtKdTree> t(new RtKdTree()); t->n_tris = n_tris; t->kind = p.kind;
        if (p.kind == RT_ACCEL_GRID) {
            build_grid(tri_verts, n_tris, t->grid);
            t->tree.nodes = t->grid.voxels; t->tree.leaf_refs = t->grid.refs; t->tree.max_depth = 0;
            std::memcpy(t->tree.bounds, t->grid.bounds, sizeof t->tree.bounds); t->tree.build_seconds = t->grid.build_seconds;
        } else build_kdtree(tri_verts, n_tris, p, t->tree);
        *out = t.release(); return RT_OK;
    });
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

int rt_film_bind(RtScene *s, void *device_accum, int32_t w, int32_t h) {
    if (!s || w < 1 || h < 1) return fail(RT_EINVAL, "rt_film_bind: bad argument");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));                  // a frame may still be accumulating into the film being replaced
    if (s->own_accum && s->accum) { HIPWARN(hipFree(s->accum)); s->accum = nullptr; }
    s->film_w = w; s->film_h = h;
    if (device_accum) { s->accum = static_cast<float *>(device_accum); s->own_accum = false; }
    else {
        HIPCHK(hipMalloc((void **)&s->accum, size_t(5) * w * h * sizeof(float))); s->own_accum = true;
        HIPCHK(hipMemsetAsync(s->accum, 0, size_t(5) * w * h * sizeof(float), s->stream));   // ordered before the first gather on this stream
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

// ImageFilm::WriteImage's per-pixel arithmetic (image.cpp:157-203) on the caller's DEVICE buffers: `dev_accum` = 5 planes of n floats (the
// part of a film a rank holds after a row-wise reduce-scatter), results to dev_rgb[n][3] / dev_alpha[n]; asynchronous on the scene's stream.
int rt_film_resolve_device(RtScene *s, const float *dev_accum, uint64_t n, int premultiply, float *dev_rgb, float *dev_alpha) {
    if (!s || !dev_accum || !dev_rgb || !dev_alpha) return fail(RT_EINVAL, "null argument");
    if (n == 0) return RT_OK;
    HIPCHK(hipSetDevice(s->device));
    hipLaunchKernelGGL(film_resolve_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s->stream, dev_accum, size_t(n), premultiply, dev_rgb, dev_alpha);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

// The same with the result interleaved, dev_rgba[n][4]: what ONE all-gather moves when every rank resolves its own rows.
int rt_film_resolve_device_rgba(RtScene *s, const float *dev_accum, uint64_t n, int premultiply, float *dev_rgba) {
    if (!s || !dev_accum || !dev_rgba) return fail(RT_EINVAL, "null argument");
    if (reinterpret_cast<uintptr_t>(dev_rgba) % 16u) return fail(RT_EINVAL, "rt_film_resolve_device_rgba: the output must be 16-byte aligned");
    if (n == 0) return RT_OK;
    HIPCHK(hipSetDevice(s->device));
    hipLaunchKernelGGL(film_resolve_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s->stream, dev_accum, size_t(n), premultiply, dev_rgba, (float *)nullptr);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

// N > 1 film merge, send side (the reference: every cropwindow process writes its own EXR, tools/exrassemble.cpp:42-75 adds them up): the rank's
// full-frame 5-plane film `dev_accum` (h rows of w) re-laid as `world` parts of `rows` film rows each, part r = [5][rows][w], rows beyond h zero;
// world * rows >= h.  dev_parts = world * 5 * rows * w floats.  Asynchronous on the scene's stream.
int rt_film_pack_parts(RtScene *s, const float *dev_accum, int32_t w, int32_t h, int32_t world, int32_t rows, float *dev_parts) {
    if (!s || !dev_accum || !dev_parts) return fail(RT_EINVAL, "null argument");
    if (w <= 0 || h <= 0 || world <= 0 || rows <= 0 || int64_t(world) * rows < h) return fail(RT_EINVAL, "rt_film_pack_parts: world * rows must cover the film's rows");
    HIPCHK(hipSetDevice(s->device));
    const size_t n_out = size_t(world) * 5u * size_t(rows) * size_t(w);
    if ((n_out + 255) / 256 > 0x7fffffffull) return fail(RT_EINVAL, "rt_film_pack_parts: film too large");
    hipLaunchKernelGGL(film_pack_parts_kernel, dim3(unsigned((n_out + 255) / 256)), dim3(256), 0, s->stream, dev_accum, w, h, rows, n_out, dev_parts);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

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
    const int grx = int(std::floor(fr.fxw + 0.5f)), gry = int(std::floor(fr.fyw + 0.5f));   // reach of a sample pixel: |x - sx| <= w + .5
    const size_t col_bytes = size_t(fr.spp * 2 + 1) * sizeof(float4);
    size_t lds_kb = 40;                                       // 3 workgroups per CU (measured 60 KB: 5.6 ms, 40 KB: 5.3 ms on C2)
    if (const char *e = knob("PBRT_HIP_GATHER_LDS_KB")) lds_kb = size_t(std::max(4, std::atoi(e)));
    int cols = int((lds_kb << 10) / col_bytes);
    if (fr.x_pixel_start + fr.x_pixel_count > 32767 || fr.y_pixel_start + fr.y_pixel_count > 32767 || fr.x_pixel_start < -32768 || fr.y_pixel_start < -32768)
        return fail(RT_EINVAL, "rt_render: film coordinates beyond 32767 (the gather packs sample footprints as int16)");
    if (cols > 16 + 2 * grx) cols = 16 + 2 * grx;
    if (cols > 256) cols = 256;                               // one thread per column resolves the record addresses of a chunk
    if (!(fr.fxw > 0.f) || !(fr.fyw > 0.f)) return fail(RT_EINVAL, "rt_render: filter widths must be positive");
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
    // ---- which film gather: film_slot_kernel for filters that reach 1 or 2 pixels either side (box .. gaussian at their default widths) and whose
    // staged sample row fits LDS; film_march_kernel for up to 3 rows; the staged gather for wider ones.  PBRT_HIP_GATHER=slot|march|staged forces one (tests).
    const bool slot_ok = grx == gry && (grx == 1 || grx == 2);
    const int slot_ncs = slot_ok ? 64 / (2 * gry + 1) + 2 * grx : 0;
    const size_t slot_lds = size_t(slot_ncs) * size_t(fr.spp + 1) * 24 + 4096 + size_t(slot_ncs) * 8 + 16;
    int which = (slot_ok && slot_lds <= 64 * 1024) ? 2 : gry <= 3 ? 1 : 0;
    if (const char *ge = knob("PBRT_HIP_GATHER")) {
        which = !std::strcmp(ge, "slot") ? 2 : !std::strcmp(ge, "march") ? 1 : !std::strcmp(ge, "staged") ? 0 : -1;
        if (which < 0) return fail(RT_EINVAL, "PBRT_HIP_GATHER: slot, march or staged");
        if (which == 2 && !(slot_ok && slot_lds <= 64 * 1024)) return fail(RT_EINVAL, "PBRT_HIP_GATHER=slot: needs equal filter reaches of 1 or 2 pixels and a sample row that fits 64 KB of LDS");
        if (which == 1 && gry > 3) return fail(RT_EINVAL, "PBRT_HIP_GATHER=march: the filter reaches more than 3 rows");
    }
    if (which == 0 && cols < 1) return fail(RT_EINVAL, "rt_render: more samples per pixel than the staged film gather holds in LDS (max ~1270; filters that reach at most 3 rows take film_march_kernel, which has no limit)");
    int rows = 0;
    if (const char *e = knob("PBRT_HIP_GATHER_ROWS")) rows = std::max(1, std::atoi(e));
    auto launch_gather = [&](const DevFrame *dfr, int row0, int row_end) -> int {      // ImageFilm::AddSample for film rows [row0, row_end), on the caller's stream
        const int nrows = row_end - row0;
        if (nrows <= 0) return RT_OK;
        if (which == 2) {
            const int nc = 64 / (2 * gry + 1);
            const unsigned nbx = unsigned((fr.x_pixel_count + nc - 1) / nc);
            int r = rows;
            if (!r) {                                         // strip height: 16 rows measured best or equal on every frame size, sample count and shard count
                r = 16;                                       // (profiles/r03_gather_rows.txt: taller = fewer waves, shorter = more halo rows); small films: 8
                if (size_t(nbx) * size_t((fr.y_pixel_count + r - 1) / r) < size_t(4) * size_t(std::max(1, s->n_cus))) r = 8;
            }
            const unsigned gb = nbx * unsigned((nrows + r - 1) / r);
            const int per_lane = (slot_ncs * fr.spp + 63) / 64;           // records a lane stages per sample row: the lookahead covers them up to RT_SLOT_PF
            auto k = grx == 1 ? (per_lane <= 4 ? film_slot_kernel<1, 1, 4> : film_slot_kernel<1, 1, RT_SLOT_PF>)
                              : (per_lane <= 4 ? film_slot_kernel<2, 2, 4> : film_slot_kernel<2, 2, RT_SLOT_PF>);
            hipLaunchKernelGGL(k, dim3(gb), dim3(64), slot_lds, s->stream, dfr, r, row0, row_end);
        } else if (which == 1) {
            const unsigned nbx = unsigned((fr.x_pixel_count + 63) / 64);
            int r = rows;
            if (!r) {                                         // strip height: the record re-reads shrink with it, the waves in flight too
                r = 32;
                while (r > 4 && size_t(nbx) * size_t((fr.y_pixel_count + r - 1) / r) < size_t(16) * size_t(std::max(1, s->n_cus))) r /= 2;
            }
            const unsigned gb = nbx * unsigned((nrows + r - 1) / r);
            auto k = gry <= 1 ? film_march_kernel<1> : gry == 2 ? film_march_kernel<2> : film_march_kernel<3>;
            hipLaunchKernelGGL(k, dim3(gb), dim3(64), 0, s->stream, dfr, grx, gry, r, row0, row_end);
        } else {
            const unsigned gb = unsigned((fr.x_pixel_count + 15) / 16) * unsigned((fr.y_pixel_count + 15) / 16);
            const size_t lds_bytes = size_t(cols) * col_bytes + size_t(cols) * sizeof(unsigned long long) + 16;
            hipLaunchKernelGGL(film_gather_kernel, dim3(gb), dim3(256), lds_bytes, s->stream, dfr, grx, gry, cols);
        }
        HIPCHK(hipGetLastError());
        return RT_OK;
    };
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
    if (!skip_film) { rc = launch_gather((const DevFrame *)s->dev_frame, 0, fr.y_pixel_count); if (rc) return rc; }
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

// The radiance of every camera sample of the last rt_render, before filtering: what Scene::Render's loop hands to
// Film::AddSample (scene.cpp:76), in the sampler's order (shard-local work order).  8 floats per sample.
int rt_samples_read(RtScene *s, uint64_t first, uint64_t count, float *out) {
    if (!s || !out) return fail(RT_EINVAL, "null argument");
    if (!s->samples || first > s->samples_last || count > s->samples_last - first) return fail(RT_ESTATE, "rt_samples_read: no frame rendered / range beyond the last frame");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (count == 0) return RT_OK;
    float4 *tmp = nullptr;
    HIPCHK(hipMalloc((void **)&tmp, size_t(count) * 2 * sizeof(float4)));
    hipLaunchKernelGGL(samples_unpack_kernel, dim3(unsigned((count + 255) / 256)), dim3(256), 0, s->stream, (const float4 *)s->samples,
                       (unsigned long long)first, (unsigned long long)count, s->samples_spp, tmp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess) e = hipMemcpy(out, tmp, size_t(count) * 2 * sizeof(float4), hipMemcpyDeviceToHost);
    HIPWARN(hipFree(tmp));
    if (e != hipSuccess) return fail(RT_EDEVICE, std::string("rt_samples_read: ") + hipGetErrorString(e));
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
