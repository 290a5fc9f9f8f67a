// rt_film.hip -- ImageFilm on the device: the three film-gather kernels (ImageFilm::AddSample, film/image.cpp:103-142, without atomics and in the reference's
// sample order), the resolve / pack kernels (ImageFilm::WriteImage's arithmetic, image.cpp:157-203), and the rt_film_* / rt_samples_read entry points of the C ABI.
#include "rt_host.h"

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

}  // namespace rt

// ---- the film gather of a frame: validation + kernel choice (called by rt_render before anything is launched), then the launch
int film_gather_plan(RtScene *s, const DevFrame &fr, FilmGather &g) {
    (void)s;
    const int grx = g.grx = int(std::floor(fr.fxw + 0.5f)), gry = g.gry = int(std::floor(fr.fyw + 0.5f));   // reach of a sample pixel: |x - sx| <= w + .5
    const size_t col_bytes = g.col_bytes = size_t(fr.spp * 2 + 1) * sizeof(float4);
    size_t lds_kb = 40;                                       // 3 workgroups per CU (measured 60 KB: 5.6 ms, 40 KB: 5.3 ms on C2)
    if (const char *e = knob("PBRT_HIP_GATHER_LDS_KB")) lds_kb = size_t(std::max(4, std::atoi(e)));
    int cols = int((lds_kb << 10) / col_bytes);
    if (fr.x_pixel_start + fr.x_pixel_count > 32767 || fr.y_pixel_start + fr.y_pixel_count > 32767 || fr.x_pixel_start < -32768 || fr.y_pixel_start < -32768)
        return fail(RT_EINVAL, "rt_render: film coordinates beyond 32767 (the gather packs sample footprints as int16)");
    if (cols > 16 + 2 * grx) cols = 16 + 2 * grx;
    if (cols > 256) cols = 256;                               // one thread per column resolves the record addresses of a chunk
    if (!(fr.fxw > 0.f) || !(fr.fyw > 0.f)) return fail(RT_EINVAL, "rt_render: filter widths must be positive");
    g.cols = cols;
    // ---- which film gather: film_slot_kernel for filters that reach 1 or 2 pixels either side (box .. gaussian at their default widths) and whose
    // staged sample row fits LDS; film_march_kernel for up to 3 rows; the staged gather for wider ones.  PBRT_HIP_GATHER=slot|march|staged forces one (tests).
    const bool slot_ok = grx == gry && (grx == 1 || grx == 2);
    const int slot_ncs = g.slot_ncs = slot_ok ? 64 / (2 * gry + 1) + 2 * grx : 0;
    const size_t slot_lds = g.slot_lds = size_t(slot_ncs) * size_t(fr.spp + 1) * 24 + 4096 + size_t(slot_ncs) * 8 + 16;
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
    g.which = which; g.rows = rows;
    return RT_OK;
}
int film_gather_launch(RtScene *s, const DevFrame &fr, const FilmGather &g, const DevFrame *dfr, int row0, int row_end) {      // ImageFilm::AddSample for film rows [row0, row_end), on the scene's stream
    const int which = g.which, grx = g.grx, gry = g.gry, cols = g.cols, rows = g.rows, slot_ncs = g.slot_ncs;
    const size_t slot_lds = g.slot_lds, col_bytes = g.col_bytes;
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
}

extern "C" {

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

}  // extern "C"
