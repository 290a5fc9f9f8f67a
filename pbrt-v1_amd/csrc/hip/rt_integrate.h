// rt_integrate.h -- the per-lane path state machine of the persistent wavefront renderer.
//
// pbrt-v1 evaluates a camera sample with recursive C++ calls (Scene::Render scene.cpp:42-84 ->
// SurfaceIntegrator::Li -> EstimateDirect -> Scene::Intersect...).  Here each lane of a 64-wide wavefront
// owns one camera sample at a time and runs it as an explicit state machine whose ONLY blocking
// operation is "trace the ray I just set up"; all lanes share one traversal loop (rt_render_kernel.h), and a
// lane whose sample finishes immediately fetches the next one (persistent threads + ray regeneration).
// Whitted / DirectLighting recursion (whitted.cpp:82-137, directlighting.cpp:127-183) is flattened to
// explicit frames that keep the reference's evaluation order, so partial sums are formed in the same
// order as the recursion forms them.
//
// RNG: draw #c of camera sample #n is pcg(c + pcg(n + seed*K)) (rt_math.h); the order in which the
// reference consumes draws per sample is reproduced exactly.
#pragma once
#include "rt_shade.h"

namespace rt {

enum Stage {
    ST_FETCH = 0, ST_VERTEX, ST_DIRECT_NEXT, ST_SHADOW_DONE, ST_MIS_DONE, ST_ED_BSDF, ST_ED_DONE,
    ST_BOUNCE, ST_SPECULAR, ST_SPEC_TRANS, ST_RETURN, ST_VOL_BEGIN, ST_VOL_STEP, ST_POP, ST_FINISH, ST_EXIT
};

struct Rng {
    uint32_t base, ctr;
    RT_DEV float next_float() { return rng_f32(base, ctr++); }
    RT_DEV uint32_t next_u32() { return rng_u32(base, ctr++); }
};

#define RT_FRAME_WORDS 25

struct Lane {
    // --- the camera sample being evaluated
    uint32_t sample_index;
    uint32_t work;          // index of this sample in the shard's work list == slot in the sample buffer
    float image_x, image_y;
    uint32_t dim_base;      // counter at which this sample's LatinHypercube block starts
    Rng rng;
    // --- radiance bookkeeping
    V3 L;                   // path: L of PathIntegrator::Li; whitted/direct: L of the current recursion frame
    V3 thr;                 // path throughput
    float alpha;
    int depth;              // pathLength (path) or rayDepth (whitted/direct)
    int fsp;                // number of suspended recursion frames
    bool specular;
    // --- current surface vertex
    Vertex v;
    // --- direct-lighting loop state
    int li, lj;             // light / sample-of-light cursor
    int cur_light;          // light of the EstimateDirect in flight
    V3 Ld;                  // EstimateDirect's Ld            (transport.cpp:127)
    V3 Ld_light;            // UniformSampleAllLights' per-light Ld (transport.cpp:42)
    V3 L_all;               // UniformSampleAllLights' L     (transport.cpp:36)
    V3 pend;                // contribution added if the ray in flight confirms it
    float bs1, bs2, bcs;    // BSDF-half sample values of the EstimateDirect in flight
    // --- DirectLighting "weighted" only (RT_INTEG_DIRECT_WEIGHTED kernels; dead in every other instantiation)
    uint32_t ord;           // ordinal of the sample's next shading point in the frame's program order (DevFrame.wt_base[work] + points so far)
    uint32_t ctr0;          // survey pass: the RNG counter at the shading point, every light's estimate starts from it
    float wt_w;             // frame pass: lightSampleWeight of the light in flight (0: uniform start-up branch)
    // --- by-vertex form of the path integrator in the megakernel (advance_pass_byv; dead in every other instantiation)
    unsigned vf;            // BV_* flags of the vertex whose rays are being traced
    V3 dM, dB;              // directions of its BSDF-sampled MIS ray and of its continuation ray (the shadow ray starts at once; all three leave v.p with mint = RAY_EPSILON)
    V3 pendS, pendM, thr_old;   // the two contributions EstimateDirect adds if its rays confirm them, and the throughput they are weighted with
    // --- control
    int stage;
    bool has_ray;
    Trav tv;
};

// ---- sampler dimensions: Sample::oneD / twoD produced by LatinHypercube (sampling.cpp:98-113) -------------
// value j, dimension d of request r.  n == 1 is the common case (a single draw, the permutation of one
// element is the identity); n > 1 replays the per-dimension swap chain backwards to find which stratum
// ends at position j, using the fact that every draw is addressable by its counter.
RT_DEV float lhs_value(const Lane &ln, const DimReq &r, int j, int d) {
    const uint32_t fb = ln.dim_base + r.f_base, ub = ln.dim_base + r.u_base;
    if (r.n == 1) return (0 + rng_f32(ln.rng.base, fb + d)) * 1.f;
    int pos = j;
    const int n = r.n;
    for (int k = n - 1; k >= 0; --k) {
        int other = int(rng_u32(ln.rng.base, ub + d * n + k) % uint32_t(n));
        if (pos == k) pos = other; else if (pos == other) pos = k;
    }
    float delta = 1.f / n;
    return (pos + rng_f32(ln.rng.base, fb + pos * r.dims + d)) * delta;
}

// (0,2)-sequence generators of core/sampling.h:137-151
RT_DEV float van_der_corput(uint32_t n, uint32_t scramble) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
    n ^= scramble;
    return (float)n / (float)0x100000000LL;
}
RT_DEV float sobol2(uint32_t n, uint32_t scramble) {
    for (uint32_t v = 1u << 31; n != 0; n >>= 1, v ^= v >> 1)
        if (n & 0x1) scramble ^= v;
    return (float)scramble / (float)0x100000000LL;
}
// Shuffle(samp, count, dims) (sampling.cpp:91-97) replayed backwards: which original element ends at `pos`.
// Its `count` RandomUInt() draws sit at counters off .. off+count-1 of the stream `base`.
RT_DEV int shuffle_origin(uint32_t base, uint32_t off, int count, int pos) {
    for (int k = count - 1; k >= 0; --k) {
        const int other = int(rng_u32(base, off + k) % uint32_t(count));
        if (pos == k) pos = other; else if (pos == other) pos = k;
    }
    return pos;
}
// key under which a pixel's sample set is generated: the key of its first sample (the keyed wrapper re-keys on
// every GetNextSample), except pixel 0 of the stratified / random samplers, whose constructor draws it.
RT_DEV uint32_t pixel_stream(const DevFrame &fr, const Lane &ln) {
    const uint32_t n0 = ln.sample_index - (ln.work % uint32_t(fr.spp));
    if (n0 == 0 && fr.sampler != RT_SAMPLER_LOWDISCREPANCY) return rng_base(0xFFFFFFFFu, fr.seed);
    return rng_base(n0, fr.seed);
}
// Value j, dimension d of one Sample::oneD/twoD request for the current camera sample, for the three samplers:
//   stratified      LatinHypercube per sample (sampling.cpp:98-113)                        -> lhs_value
//   random          one RandomFloat() per value, in request order (random.cpp:107-112)
//   lowdiscrepancy  per-pixel scrambled (0,2)-sequence tables, shuffled within each sample's block and then across
//                   the pixel's samples (lowdiscrepancy.cpp:93-104, sampling.h:152-174); every draw addressable
RT_DEV float dim_value(const DevFrame &fr, const Lane &ln, const DimReq &r, int j, int d) {
    if (fr.sampler == RT_SAMPLER_STRATIFIED) return lhs_value(ln, r, j, d);
    if (fr.sampler == RT_SAMPLER_RANDOM) return rng_f32(ln.rng.base, ln.dim_base + r.f_base + j * r.dims + d);
    const uint32_t pb = pixel_stream(fr, ln);
    const int P = fr.spp, n = r.n, s = int(ln.work % uint32_t(P));
    const uint32_t nscr = r.dims;                                    // 1 or 2 scramble words lead the block
    const int blk = shuffle_origin(pb, r.f_base + nscr + uint32_t(n) * P, P, s);
    const int jb = (n == 1) ? 0 : shuffle_origin(pb, r.f_base + nscr + uint32_t(blk) * n, n, j);
    const uint32_t idx = uint32_t(blk) * n + jb;
    if (r.dims == 1) return van_der_corput(idx, rng_u32(pb, r.f_base));
    return d == 0 ? van_der_corput(idx, rng_u32(pb, r.f_base)) : sobol2(idx, rng_u32(pb, r.f_base + 1));
}

// ---- Scene::Render's radiance sanity check (scene.cpp:60-74); the sample's value is parked in the per-shard
// sample buffer (32 B: L.rgb, alpha, imageX, imageY) and splatted by film_gather_kernel, which replays
// ImageFilm::AddSample (image.cpp:103-142) per pixel in the reference's sample order.  Compared with
// atomically splatting each sample into (2w+1)^2 pixels x 5 planes this removes ~100 L2 atomics per sample
// (59 % of the frame time on the Cornell path-tracing config) and makes the film deterministic.
#ifndef RT_SAMPLE_NT
#define RT_SAMPLE_NT 1
#endif
RT_DEV void sample_write(const DevFrame &fr, const Lane &ln, V3 Ls, float alpha, unsigned &bad) {
    const float y = lum_y(Ls);
    if (Ls.x != Ls.x || Ls.y != Ls.y || Ls.z != Ls.z) { Ls = mk3(0.f); ++bad; }
    else if (y < -1e-5) { Ls = mk3(0.f); ++bad; }
    else if (isinf(y)) { Ls = mk3(0.f); ++bad; }
    const uint32_t lp = ln.work / uint32_t(fr.spp);
    float4 RT_G *rec = RT_GPTR(float4, fr.samples) + sample_slot(lp, ln.work - lp * uint32_t(fr.spp), fr.spp);
#if RT_SAMPLE_NT
    // written once, read once by the film gather after the kernel: keep the records out of the way of the tree's lines in L2 (non-temporal stores)
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const nt_f4 r0 = {Ls.x, Ls.y, Ls.z, alpha}, r1 = {ln.image_x, ln.image_y, 0.f, 0.f};
    __builtin_nontemporal_store(r0, (nt_f4 RT_G *)rec);
    __builtin_nontemporal_store(r1, (nt_f4 RT_G *)(rec + RT_SAMPLE_XY));
#else
    rec[0] = make_float4(Ls.x, Ls.y, Ls.z, alpha);
    rec[RT_SAMPLE_XY] = make_float4(ln.image_x, ln.image_y, 0.f, 0.f);
#endif
}

// ---- camera sample -> camera ray -----------------------------------------------------------------------
// StratifiedSampler (samplers/stratified.cpp:51-131, sampling.cpp:72-97): image position of sample s of
// pixel (px,py); lens position only when the camera has a lens (needs the Shuffle replay).
RT_DEV void stratified_camera_sample(const DevFrame &fr, uint32_t pixel_index, int s, int px, int py,
                                     uint32_t pix_base, float &ix, float &iy, float &lu, float &lv, bool need_lens) {
    const int n = fr.spp;
    const int sx = s % fr.xs, sy = s / fr.xs;
    const float dx = 1.f / fr.xs, dy = 1.f / fr.ys;
    float jx = 0.5f, jy = 0.5f;
    if (fr.jitter) { jx = rng_f32(pix_base, 2 * s); jy = rng_f32(pix_base, 2 * s + 1); }
    ix = (sx + jx) * dx; iy = (sy + jy) * dy;
    ix += px; iy += py;
    lu = lv = 0.5f;
    if (need_lens) {
        // Shuffle(lensSamples, n, 2): draws start after the strata draws (5n when jittered, else 0)
        const uint32_t ub = fr.jitter ? 5u * n : 0u;
        int pos = s;
        for (int k = n - 1; k >= 0; --k) {
            int other = int(rng_u32(pix_base, ub + k) % uint32_t(n));
            if (pos == k) pos = other; else if (pos == other) pos = k;
        }
        const int lx = pos % fr.xs, ly = pos / fr.xs;
        float kx = 0.5f, ky = 0.5f;
        if (fr.jitter) { kx = rng_f32(pix_base, 2 * n + 2 * pos); ky = rng_f32(pix_base, 2 * n + 2 * pos + 1); }
        lu = (lx + kx) * dx; lv = (ly + ky) * dy;
    }
    (void)pixel_index;
}

// {Perspective,Ortho,Environment}Camera::GenerateRay (cameras/perspective.cpp:51-82, orthographic.cpp:48-79,
// environment.cpp:47-61)
RT_DEV Ray camera_ray(const RtCamera &cam, float ix, float iy, float lensU, float lensV) {
    Ray r;
    if (cam.type == RT_CAMERA_ENVIRONMENT) {
        r.o = xform_point(cam.camera_to_world, mk3(0.f, 0.f, 0.f));               // rayOrigin = CameraToWorld(Point(0,0,0))
        const float theta = RT_PI * iy / cam.y_res;                               // M_PI is a float (pbrt.h:204); int -> float promotion
        const float phi = 2 * RT_PI * ix / cam.x_res;
        const V3 dir = mk3(sinf(theta) * cosf(phi), cosf(theta), sinf(theta) * sinf(phi));
        r.d = xform_vector(cam.camera_to_world, dir);
        r.mint = cam.hither; r.maxt = cam.yon;
        return r;
    }
    V3 Pcamera = xform_point(cam.raster_to_camera, mk3(ix, iy, 0.f));
    r.o = Pcamera;
    r.d = cam.type == RT_CAMERA_ORTHOGRAPHIC ? mk3(0.f, 0.f, 1.f) : Pcamera;
    if (cam.lens_radius > 0.f) {
        float lu, lv; concentric_disk(lensU, lensV, lu, lv);
        lu *= cam.lens_radius; lv *= cam.lens_radius;
        float ft = (cam.focal_distance - cam.hither) / r.d.z;
        V3 Pfocus = r.o + r.d * ft;
        r.o.x += lu * (cam.focal_distance - cam.hither) / cam.focal_distance;
        r.o.y += lv * (cam.focal_distance - cam.hither) / cam.focal_distance;
        r.d = Pfocus - r.o;
    }
    r.d = normalize3(r.d);
    r.mint = 0.f;
    r.maxt = cam.type == RT_CAMERA_ORTHOGRAPHIC ? cam.yon - cam.hither : (cam.yon - cam.hither) / r.d.z;
    r.o = xform_point(cam.camera_to_world, r.o);
    r.d = xform_vector(cam.camera_to_world, r.d);
    return r;
}

// map a work index of this shard to (pixel, sample-in-pixel); false if it falls off the image
RT_DEV bool work_to_sample(const DevFrame &fr, unsigned long long w, unsigned long long &pixel, int &s) {
    const unsigned long long per_tile = (unsigned long long)fr.tile_pixels * fr.spp;
    // every frame of practical size has fewer than 2^32 samples: 32-bit divisions (a 64-bit division is ~150 VALU instructions
    // on gfx950, and this runs in the sparsely populated fetch path)
    if (fr.tile_w > 0) {                                    // 2-D tiles: a tile is a tile_w x tile_h block of the sample extent; the pixels of
        // the border tiles that fall off the extent are skipped.  make_frame guarantees n_tiles * tile_pixels * spp < 2^32: all 32-bit
        const unsigned w32 = unsigned(w), pt = unsigned(per_tile);
        const unsigned lt = w32 / pt, rem = w32 - lt * pt;
        const unsigned tile = lt * unsigned(fr.shard_count) + unsigned(fr.shard_index);
        const unsigned q = rem / unsigned(fr.spp);
        s = int(rem - q * unsigned(fr.spp));
        const unsigned ty = tile / unsigned(fr.tiles_x), tx = tile - ty * unsigned(fr.tiles_x);
        const unsigned qy = q / unsigned(fr.tile_w), qx = q - qy * unsigned(fr.tile_w);
        const unsigned px = tx * unsigned(fr.tile_w) + qx, py = ty * unsigned(fr.tile_h) + qy;
        const unsigned ew = unsigned(fr.x_end - fr.x_start), eh = unsigned(fr.y_end - fr.y_start);
        pixel = (unsigned long long)py * ew + px;
        return px < ew && py < eh;
    }
    if (fr.total_work <= 0xffffffffull && per_tile <= 0xffffffffull) {
        const unsigned w32 = unsigned(w), pt = unsigned(per_tile);
        const unsigned lt = w32 / pt, rem = w32 - lt * pt;
        const unsigned long long tile = (unsigned long long)lt * unsigned(fr.shard_count) + unsigned(fr.shard_index);
        const unsigned q = rem / unsigned(fr.spp);
        pixel = tile * unsigned(fr.tile_pixels) + q;
        s = int(rem - q * unsigned(fr.spp));
        return pixel < fr.total_pixels;
    }
    const unsigned long long lt = w / per_tile, rem = w % per_tile;
    const unsigned long long tile = lt * fr.shard_count + fr.shard_index;
    pixel = tile * fr.tile_pixels + rem / fr.spp;
    s = int(rem % fr.spp);
    return pixel < fr.total_pixels;
}

// Single shard, megakernel (round 4): the ORDER in which the work counter hands out the samples -- square tiles of mega_tile x mega_tile pixels,
// row-major over the sample extent, clipped at its right and bottom edge (no padding, so no dropped items), a tile's pixels in scanline order, a
// pixel's samples consecutively.  Everything else keeps addressing a sample by its scanline index pixel * spp + s (ln.work, the sample buffer,
// rt_samples_read): only which lanes render which samples at the same time changes -- the waves resident on an XCD then work on a compact block of the
// image instead of a strip one pixel high (their rays meet the same part of the tree: L2).
RT_DEV void tile_order_to_sample(const DevFrame &fr, unsigned w, unsigned &pixel, int &s) {
    const unsigned spp = unsigned(fr.spp), T = unsigned(fr.mega_tile);
    const unsigned W = unsigned(fr.x_end - fr.x_start), H = unsigned(fr.y_end - fr.y_start);
    const unsigned wp = w / spp; s = int(w - wp * spp);
    const unsigned row_px = W * T, ty = wp / row_px, rem = wp - ty * row_px;
    const unsigned h_t = min(T, H - ty * T), full = T * h_t, tiles_x = (W + T - 1u) / T;
    unsigned tx = rem / full; tx = tx < tiles_x ? tx : tiles_x - 1u;
    const unsigned rem2 = rem - tx * full, w_t = min(T, W - tx * T);
    const unsigned qy = rem2 / w_t, qx = rem2 - qy * w_t;
    pixel = (ty * T + qy) * W + tx * T + qx;
}

RT_DEV void setup_sample(const DevScene &sc, const DevFrame &fr, Lane &ln, unsigned long long pixel, int s, Ray &ray) {
    const int w = fr.x_end - fr.x_start;
    int px, py;
    if (fr.total_pixels <= 0xffffffffull) { const unsigned row = unsigned(pixel) / unsigned(w); px = fr.x_start + int(unsigned(pixel) - row * unsigned(w)); py = fr.y_start + int(row); }
    else { px = fr.x_start + int(pixel % w); py = fr.y_start + int(pixel / w); }
    const uint32_t n0 = uint32_t(pixel * fr.spp);
    ln.sample_index = n0 + uint32_t(s);
    ln.rng.base = rng_base(ln.sample_index, fr.seed);
    // the first pixel's strata are drawn by the sampler's constructor, before any sample key exists
    float lu = 0.5f, lv = 0.5f;
    const bool need_lens = sc.cam.lens_radius > 0.f;
    if (fr.sampler == RT_SAMPLER_STRATIFIED) {
        // the first pixel's strata are drawn by the sampler's constructor, before any sample key exists
        const uint32_t pix_base = (pixel == 0) ? rng_base(0xFFFFFFFFu, fr.seed) : rng_base(n0, fr.seed);
        ln.dim_base = (s == 0 && pixel != 0) ? fr.pixgen_draws : 0u;
        stratified_camera_sample(fr, uint32_t(pixel), s, px, py, pix_base, ln.image_x, ln.image_y, lu, lv, need_lens);
    } else if (fr.sampler == RT_SAMPLER_RANDOM) {                       // random.cpp:45-106
        const uint32_t pix_base = (pixel == 0) ? rng_base(0xFFFFFFFFu, fr.seed) : rng_base(n0, fr.seed);
        ln.dim_base = (s == 0 && pixel != 0) ? fr.pixgen_draws : 0u;
        ln.image_x = rng_f32(pix_base, 2 * s); ln.image_y = rng_f32(pix_base, 2 * s + 1);
        ln.image_x += px; ln.image_y += py;
        if (need_lens) { lu = rng_f32(pix_base, 2 * fr.spp + 2 * s); lv = rng_f32(pix_base, 2 * fr.spp + 2 * s + 1); }
    } else {                                                            // lowdiscrepancy.cpp:76-128
        const uint32_t pix_base = rng_base(n0, fr.seed);
        const uint32_t P = uint32_t(fr.spp);
        ln.dim_base = (s == 0) ? fr.pixgen_draws : 0u;                  // every pixel is generated inside GetNextSample
        const int pos = shuffle_origin(pix_base, 2 + P, int(P), s);     // image block: 2 scrambles, P no-op draws, P-shuffle
        ln.image_x = px + van_der_corput(uint32_t(pos), rng_u32(pix_base, 0));
        ln.image_y = py + sobol2(uint32_t(pos), rng_u32(pix_base, 1));
        if (need_lens) {
            const uint32_t lb = 2 + 2 * P;
            const int lp = shuffle_origin(pix_base, lb + 2 + P, int(P), s);
            lu = van_der_corput(uint32_t(lp), rng_u32(pix_base, lb)); lv = sobol2(uint32_t(lp), rng_u32(pix_base, lb + 1));
        }
    }
    ln.rng.ctr = ln.dim_base + ((fr.sampler == RT_SAMPLER_LOWDISCREPANCY) ? 0u : fr.lhs_total);
    ray = camera_ray(sc.cam, ln.image_x, ln.image_y, lu, lv);
    // scene.cpp:47-53 generates the differential's offset rays with ++ / -- on sample->imageX / imageY: Film::AddSample later sees
    // (x + 1) - 1 in float arithmetic, not x (found by the reference-side binding test, tests/test_boundary.py)
    ln.image_x = (ln.image_x + 1.f) - 1.f; ln.image_y = (ln.image_y + 1.f) - 1.f;
}

// ---- recursion frames (whitted / directlighting) ---------------------------------------------------------
RT_DEV float RT_G *frame_ptr(const DevFrame &fr, int frame, unsigned gtid) {
    return RT_GPTR(float, fr.frames) + (size_t(frame) * RT_FRAME_WORDS) * fr.n_threads + gtid;
}
template <bool EXT>
RT_DEV void frame_push(const DevFrame &fr, Lane &ln, unsigned gtid, V3 f, float absdot, int after) {
    float RT_G *q = frame_ptr(fr, ln.fsp, gtid); const size_t st = fr.n_threads;
    q[0 * st] = ln.L.x; q[1 * st] = ln.L.y; q[2 * st] = ln.L.z;
    q[3 * st] = f.x; q[4 * st] = f.y; q[5 * st] = f.z; q[6 * st] = absdot;
    q[7 * st] = __int_as_float(after); q[8 * st] = __int_as_float(ln.depth);
    q[9 * st] = ln.v.p.x; q[10 * st] = ln.v.p.y; q[11 * st] = ln.v.p.z;
    q[12 * st] = ln.v.nn.x; q[13 * st] = ln.v.nn.y; q[14 * st] = ln.v.nn.z;
    q[15 * st] = ln.v.sn.x; q[16 * st] = ln.v.sn.y; q[17 * st] = ln.v.sn.z;
    q[18 * st] = ln.v.wo.x; q[19 * st] = ln.v.wo.y; q[20 * st] = ln.v.wo.z;
    q[21 * st] = __int_as_float(ln.v.mat);
    if (EXT) { q[22 * st] = ln.v.ng.x; q[23 * st] = ln.v.ng.y; q[24 * st] = ln.v.ng.z; }
    ++ln.fsp;
}
template <bool EXT>
RT_DEV int frame_pop(const DevFrame &fr, Lane &ln, unsigned gtid, V3 child) {
    --ln.fsp;
    const float RT_G *q = frame_ptr(fr, ln.fsp, gtid); const size_t st = fr.n_threads;
    V3 Lp = mk3(q[0 * st], q[1 * st], q[2 * st]);
    V3 f = mk3(q[3 * st], q[4 * st], q[5 * st]);
    float absdot = q[6 * st];
    ln.L = Lp + (child * f) * absdot;                       // L += scene->Li(rd) * f * AbsDot(wi, n)
    int after = __float_as_int(q[7 * st]);
    ln.depth = __float_as_int(q[8 * st]);
    ln.v.p = mk3(q[9 * st], q[10 * st], q[11 * st]);
    ln.v.nn = mk3(q[12 * st], q[13 * st], q[14 * st]);
    ln.v.sn = mk3(q[15 * st], q[16 * st], q[17 * st]);
    ln.v.tn = cross3(ln.v.nn, ln.v.sn);
    ln.v.wo = mk3(q[18 * st], q[19 * st], q[20 * st]);
    ln.v.mat = __float_as_int(q[21 * st]);
    if (EXT) ln.v.ng = mk3(q[22 * st], q[23 * st], q[24 * st]);
    return after;
}

// DEFER (queue pipeline, rt_pipeline.h): the ray is only recorded; the trace kernel starts the traversal
template <bool DEFER = false>
RT_DEV void launch_ray(Lane &ln, const DevScene &sc, V3 o, V3 d, float mint, float maxt, bool any, int next_stage) {
    Ray r; r.o = o; r.d = d; r.mint = mint; r.maxt = maxt;
    if (DEFER) { ln.tv.o = o; ln.tv.d = d; ln.tv.mint = mint; ln.tv.maxt = maxt; ln.tv.any = any; ln.tv.hit_prim = -1; ln.tv.b1 = 0.f; ln.tv.b2 = 0.f; }
    else if (sc.accel_kind == RT_ACCEL_GRID) grid_begin(ln.tv, sc, r, any); else trav_begin(ln.tv, sc, r, any);
    ln.has_ray = true;
    ln.stage = next_stage;
}

// ---- participating medium: HomogeneousVolume (volumes/homogeneous.cpp:27-74) and the volume integrators --------
// Per-thread scratch in HBM (only allocated / touched when the scene has a volume):
//   vol_rays [level][8][thread]   the ray of the Scene::Li invocation at recursion level `fsp` (o, d, mint, maxt)
//   vol_state[13][thread]         suspended ray-march state of SingleScattering::Li while a shadow ray is traced
//   vol_samp [3*Nmax][thread]     its LatinHypercube(samp, N, 3) table (single.cpp:76-77)
RT_DEV float RT_G *vol_ray_ptr(const DevFrame &fr, int level, unsigned gtid) { return RT_GPTR(float, fr.vol_rays) + size_t(level) * 8 * fr.n_threads + gtid; }
RT_DEV void vol_store_ray(const DevFrame &fr, int level, unsigned gtid, const Ray &r) {
    float RT_G *q = vol_ray_ptr(fr, level, gtid); const size_t st = fr.n_threads;
    q[0] = r.o.x; q[st] = r.o.y; q[2 * st] = r.o.z; q[3 * st] = r.d.x; q[4 * st] = r.d.y; q[5 * st] = r.d.z; q[6 * st] = r.mint; q[7 * st] = r.maxt;
}
RT_DEV Ray vol_load_ray(const DevFrame &fr, int level, unsigned gtid) {
    const float RT_G *q = vol_ray_ptr(fr, level, gtid); const size_t st = fr.n_threads;
    Ray r; r.o = mk3(q[0], q[st], q[2 * st]); r.d = mk3(q[3 * st], q[4 * st], q[5 * st]); r.mint = q[6 * st]; r.maxt = q[7 * st];
    return r;
}
// HomogeneousVolume::IntersectP :43-46 (BBox::IntersectP on the volume-space ray)
RT_DEV bool vol_intersect(const RtVolume &v, V3 o, V3 d, float mint, float maxt, float &t0, float &t1) {
    const V3 vo = xform_point(v.world_to_volume, o), vd = xform_vector(v.world_to_volume, d);
    float a = mint, b = maxt; bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float invRayDir = 1.f / comp(vd, i);
        float tNear = (v.p0[i] - comp(vo, i)) * invRayDir, tFar = (v.p1[i] - comp(vo, i)) * invRayDir;
        if (tNear > tFar) { float tmp = tNear; tNear = tFar; tFar = tmp; }
        a = tNear > a ? tNear : a; b = tFar < b ? tFar : b;
        if (a > b) ok = false;
    }
    t0 = a; t1 = b; return ok;
}
RT_DEV bool vol_inside(const RtVolume &v, V3 p) {
    const V3 q = xform_point(v.world_to_volume, p);
    return q.x >= v.p0[0] && q.x <= v.p1[0] && q.y >= v.p0[1] && q.y <= v.p1[1] && q.z >= v.p0[2] && q.z <= v.p1[2];
}
// exp(-Tau(ray)) with Tau = Distance(ray(t0), ray(t1)) * (sigma_a + sigma_s)   (homogeneous.cpp:63-67, emission.cpp:47-59)
RT_DEV V3 vol_transmittance(const RtVolume &v, V3 o, V3 d, float mint, float maxt) {
    float t0, t1;
    if (!vol_intersect(v, o, d, mint, maxt, t0, t1)) return mk3(1.f);
    const float dist = len3((o + d * t0) - (o + d * t1));
    const V3 tau = (mat_color(v.sigma_a) + mat_color(v.sigma_s)) * dist;
    return mk3(expf(-tau.x), expf(-tau.y), expf(-tau.z));
}
// Scene::Transmittance(ray) (scene.cpp:127-129): sample == NULL => one RandomFloat() for the (unused) offset
template <bool VOL>
RT_DEV V3 scene_transmittance(const DevScene &sc, Lane &ln, V3 o, V3 d, float mint, float maxt) {
    if (!VOL) return mk3(1.f);
    (void)ln.rng.next_float();
    return vol_transmittance(sc.vol, o, d, mint, maxt);
}

// ---- EstimateDirect (core/transport.cpp:123-194), split at its two ray casts ------------------------------
// BSDF-sampling half; returns with either a MIS ray in flight (ST_MIS_DONE) or ST_ED_DONE.
template <bool EXT, bool DEFER = false>
RT_DEV void estimate_direct_bsdf(const DevScene &sc, Lane &ln) {
    LightRef Lt = RT_LIGHT(sc, ln.cur_light);
    ln.stage = ST_ED_DONE;
    if (light_is_delta(Lt)) return;                                             // IsDeltaLight()
    MatRef m = RT_MAT(sc, ln.v.mat);
    V3 wi; float bsdfPdf; int sampled;
    V3 f = bsdf_sample_f<EXT>(m, ln.v, ln.v.wo, wi, ln.bs1, ln.bs2, ln.bcs, bsdfPdf, BX_ALL & ~BX_SPECULAR, sampled);
    if (!is_black(f) && bsdfPdf > 0.f) {
        float lightPdf = area_light_pdf<EXT>(sc, Lt, ln.v.p, wi);
        if (lightPdf > 0.f) {
            float fw = 1 * bsdfPdf, gw = 1 * lightPdf;                            // PowerHeuristic mc.h:55-59
            float weight = (fw * fw) / (fw * fw + gw * gw);
            // Li is Lemit iff the closest hit is this emitter seen from its front side; decided after the trace
            ln.pend = div_s(((f * mat_color(Lt.color)) * absdot3(wi, ln.v.nn)) * weight, bsdfPdf);
            launch_ray<DEFER>(ln, sc, ln.v.p, wi, RT_RAY_EPSILON, RT_INF, false, ST_MIS_DONE);
        }
    }
}

// DirectLighting "weighted" (rt_weighted.h): does one EstimateDirect of this light draw a random number (ShapeSet::Sample's triangle pick, shape.h:115-121)?
RT_DEV bool light_draws_rng(LightRef L) { return !light_is_delta(L) && L.quadric < 0 && L.n_tris > 1u; }
// ... and where the survey's record of the lane's current shading point starts in DevFrame::wt_rec (floats)
RT_DEV size_t weighted_record(const DevFrame &fr, const Lane &ln, int nLights) {
    if (!fr.wt_mixed) return size_t(ln.ord) * size_t(1 + 2 * nLights);
    const size_t j = ln.ord - RT_GPTR(const unsigned, fr.wt_base)[ln.work];
    return size_t(RT_GPTR(const unsigned, fr.wt_recbase)[ln.work]) + j * size_t(1 + 2 * (nLights - int(fr.wt_nd))) + size_t(fr.wt_nd) * j * (j + 1);
}

// light-sampling half
template <bool EXT, bool DEFER = false>
RT_DEV void estimate_direct_begin(const DevScene &sc, Lane &ln, int light, float ls1, float ls2) {
    ln.cur_light = light;
    ln.Ld = mk3(0.f);
    LightRef Lt = RT_LIGHT(sc, light);
    MatRef m = RT_MAT(sc, ln.v.mat);
    V3 wi, Li, sd; float lightPdf, smax;
    if (light_is_delta(Lt)) {                                                   // point.cpp:61-66, spot.cpp:80-84, distant.cpp:63-67
        Li = delta_light_sample(Lt, ln.v.p, wi, sd, smax);
        lightPdf = 1.f;
    } else {                                                                    // area.cpp:58-68
        V3 ns;
        V3 ps = area_sample_point<EXT>(sc, Lt, ln.v.p, ls1, ls2, ln.rng, ns);
        wi = normalize3(ps - ln.v.p);
        lightPdf = area_light_pdf<EXT>(sc, Lt, ln.v.p, wi);
        Li = area_L(Lt, ns, -wi);
        sd = ps - ln.v.p; smax = 1.f - RT_RAY_EPSILON;
    }
    if (lightPdf > 0.f && !is_black(Li)) {
        V3 f = bsdf_f<EXT>(m, ln.v, ln.v.wo, wi);
        if (!is_black(f)) {
            if (light_is_delta(Lt)) ln.pend = div_s((f * Li) * absdot3(wi, ln.v.nn), lightPdf);
            else {
                float bsdfPdf = bsdf_pdf<EXT>(m, ln.v, ln.v.wo, wi);
                float fw = 1 * lightPdf, gw = 1 * bsdfPdf;
                float weight = (fw * fw) / (fw * fw + gw * gw);
                ln.pend = div_s(((f * Li) * absdot3(wi, ln.v.nn)) * weight, lightPdf);
            }
            // VisibilityTester::SetSegment / SetRay light.h:78-83
            launch_ray<DEFER>(ln, sc, ln.v.p, sd, RT_RAY_EPSILON, smax, true, ST_SHADOW_DONE);
            return;
        }
    }
    ln.stage = ST_ED_BSDF;
}

// ---- the volume integrators: EmissionIntegrator::Li emission.cpp:60-95 / SingleScattering::Li single.cpp:57-116 along `ray`, then
// Scene::Li's T * Lo + Lv with the integrator's Transmittance (scene.cpp:120-126, emission.cpp:47-59).  One function for the megakernel's
// stages (state parked in HBM between calls) and the march kernel of the queue pipeline (state in registers): same operations in the same order.
// `begin`: start the march (ST_VOL_BEGIN); otherwise resume after the shadow ray of step m.i (ST_VOL_STEP: ln.tv holds its result).
// Returns true with the next step's shadow ray set up (ln.has_ray, ln.pend, ln.stage = ST_VOL_STEP), false when the level is complete
// (ln.L = T * L + Lv, ln.stage = ST_POP).  samp[(3 * a + b) * st]: the march's LatinHypercube(samp, N, 3) table (single.cpp:76-77).
struct March { int i, N; float t0, step; V3 Tr, p, Lv; float s0, s1, s2; };      // s0..s2: the sample-table row of step i + 1, requested when step i's shadow ray leaves (march kernel only)
// The head of the march: clip against the medium, step count and size, the scatter offset, the LatinHypercube table (3 N draws, then 3 N
// dependent swaps through memory: run where thousands of marches start side by side -- the shade pass of the queue pipeline, or the megakernel's
// lanes -- never inside the persistent march kernel, where one lane's table would stall its wave for ~6 N memory round trips).  N = 0: nothing to march.
RT_DEV void march_begin(const DevScene &sc, const DevFrame &fr, Lane &ln, const Ray &ray, March &m, float RT_G *samp, size_t st) {
    const RtVolume &vol = sc.vol;
    const bool single = fr.volume_integrator == RT_VOLUME_SINGLE;
    int N; float t0, t1, step; V3 p;
    if (!vol_intersect(vol, ray.o, ray.d, ray.mint, ray.maxt, t0, t1) || (t1 - t0) == 0.f) { N = 0; step = 0.f; p = ray.o; }
    else {
        N = int(ceilf((t1 - t0) / fr.step_size));
        step = (t1 - t0) / N;
        p = ray.o + ray.d * t0;
        t0 += dim_value(fr, ln, fr.one_d[fr.n1d - 1], 0, 0) * step;              // scatterSampleOffset
        if (single) {                                                          // LatinHypercube(samp, N, 3), sampling.cpp:98-113
            if (N > fr.vol_nmax) { N = fr.vol_nmax; }                           // cannot happen: vol_nmax bounds the box diagonal
            const float delta = 1.f / N;
            for (int a = 0; a < N; ++a) for (int b = 0; b < 3; ++b) samp[size_t(3 * a + b) * st] = (a + ln.rng.next_float()) * delta;
            for (int b = 0; b < 3; ++b) for (int a = 0; a < N; ++a) {
                const int other = int(ln.rng.next_u32() % uint32_t(N));
                const float tmp = samp[size_t(3 * a + b) * st]; samp[size_t(3 * a + b) * st] = samp[size_t(3 * other + b) * st]; samp[size_t(3 * other + b) * st] = tmp;
            }
        }
    }
    m.i = 0; m.N = N; m.t0 = t0; m.step = step; m.Tr = mk3(1.f); m.p = p; m.Lv = mk3(0.f);
}
// The steps.  `resume`: ln.tv holds the result of the shadow ray of step m.i.  Returns true with the next step's shadow ray set up (ln.has_ray,
// ln.pend, ln.stage = ST_VOL_STEP), false when the level is complete (ln.L = T * L + Lv, ln.stage = ST_POP).
// PREFETCH (march kernel): the next step's table row is requested together with this step's shadow ray, so the wave does not sit through an
// HBM round trip when the ray comes back (a persistent wave has nothing else to run meanwhile).
template <bool COUNT, bool EXT, bool DEFER, bool PREFETCH = false>
RT_DEV bool march_steps(const DevScene &sc, const DevFrame &fr, Lane &ln, const Ray &ray, March &m, const float RT_G *samp, size_t st, bool resume, unsigned *c_any) {
    const RtVolume &vol = sc.vol;
    const bool single = fr.volume_integrator == RT_VOLUME_SINGLE;
    int i = m.i; const int N = m.N; float t0 = m.t0; const float step = m.step; V3 Tr = m.Tr, p = m.p, Lv = m.Lv;
    const V3 w = -ray.d;
    bool have_row = false;
    if (resume) {
        if (COUNT) ++*c_any;
        if (ln.tv.hit_prim < 0)                                                // vis.Unoccluded: Ld = L * vis.Transmittance(scene)
            Lv = Lv + ln.pend * scene_transmittance<true>(sc, ln, ln.tv.o, ln.tv.d, ln.tv.mint, ln.tv.maxt);
        ++i; t0 += step;
        have_row = PREFETCH;
    }
    while (i < N) {
        const V3 pPrev = p; p = ray.o + ray.d * t0;
        (void)ln.rng.next_float();                                             // Tau's offset argument
        const V3 stepT = vol_transmittance(vol, pPrev, p - pPrev, 0.f, 1.f);
        Tr = Tr * stepT;
        if (lum_y(Tr) < 1e-3) {
            if (ln.rng.next_float() > .5f) break;
            Tr = div_s(Tr, .5f);
        }
        const bool in = vol_inside(vol, p);
        Lv = Lv + Tr * (in ? mat_color(vol.le) : mk3(0.f));
        if (single) {
            const V3 ss = in ? mat_color(vol.sigma_s) : mk3(0.f);
            const int nLights = int(sc.n_lights);
            if (!is_black(ss) && nLights > 0) {
                const float r0 = have_row ? m.s0 : samp[size_t(3 * i) * st], u1 = have_row ? m.s1 : samp[size_t(3 * i + 1) * st], u2 = have_row ? m.s2 : samp[size_t(3 * i + 2) * st];
                const int lightNum = min(int(floorf(r0 * nLights)), nLights - 1);
                LightRef Lt = RT_LIGHT(sc, lightNum);
                V3 wo, L, sd; float pdf, smax;
                if (light_is_delta(Lt)) { L = delta_light_sample(Lt, p, wo, sd, smax); pdf = 1.f; }
                else {
                    V3 ns; V3 ps = area_sample_point<EXT>(sc, Lt, p, u1, u2, ln.rng, ns);
                    wo = normalize3(ps - p); pdf = area_light_pdf<EXT>(sc, Lt, p, wo); L = area_L(Lt, ns, -wo);
                    sd = ps - p; smax = 1.f - RT_RAY_EPSILON;
                }
                if (!is_black(L) && pdf > 0.f) {
                    const float costheta = dot3(w, -wo);                       // PhaseHG volume.cpp:44-48
                    const float phase = in ? 1.f / (4.f * RT_PI) * (1.f - vol.g * vol.g) / powf(1.f + vol.g * vol.g - 2.f * vol.g * costheta, 1.5f) : 0.f;
                    ln.pend = div_s((((Tr * ss) * phase) * L) * float(nLights), pdf);
                    m.i = i; m.t0 = t0; m.Tr = Tr; m.p = p; m.Lv = Lv;
                    if (PREFETCH) {                                            // row i + 1 (row N - 1 again past the end: never used)
                        const int j = i + 1 < N ? i + 1 : i;
                        m.s0 = samp[size_t(3 * j) * st]; m.s1 = samp[size_t(3 * j + 1) * st]; m.s2 = samp[size_t(3 * j + 2) * st];
                    }
                    launch_ray<DEFER>(ln, sc, p, sd, RT_RAY_EPSILON, smax, true, ST_VOL_STEP);
                    return true;
                }
            }
        }
        ++i; t0 += step; have_row = false;
    }
    Lv = Lv * step;
    const V3 T = vol_transmittance(vol, ray.o, ray.d, ray.mint, ray.maxt);      // sample != NULL: no draw
    ln.L = T * ln.L + Lv;
    ln.stage = ST_POP;
    return false;
}

// The body of ONE stage.  Returns when the lane has a ray in flight or has changed stage.
template <bool COUNT, int INTEG_, bool VOL, bool EXT, int STAGE, bool DEFER = false, bool PARK = false>
RT_DEV void stage_body(const DevScene &sc, const DevFrame &fr, Lane &ln, unsigned gtid,
                       unsigned *c_closest, unsigned *c_any, unsigned *c_bad) {
    // RT_INTEG_DIRECT_WEIGHTED: DirectLighting with strategy "weighted", a kernel family of its own so that the all / one kernels keep their code
    constexpr bool WEIGHTED = INTEG_ == RT_INTEG_DIRECT_WEIGHTED;
    constexpr int INTEG = WEIGHTED ? int(RT_INTEGRATOR_DIRECT) : INTEG_;
    if constexpr (STAGE == ST_VERTEX) {
        if (COUNT) ++*c_closest;
        const bool hit = ln.tv.hit_prim >= 0;
        if (INTEG == RT_INTEGRATOR_PATH) {
            if (!hit) {                                                         // path.cpp:68-83: point/area lights have Le(ray)=0
                if (ln.depth == 0) ln.alpha = (ln.L.x != 0.f || ln.L.y != 0.f || ln.L.z != 0.f) ? 1.f : 0.f;
                ln.stage = ST_RETURN; return;
            }
            make_vertex<EXT>(sc, ln.tv, ln.v);
#ifdef RT_DEBUG_PIXEL
            if (int(floorf(ln.image_x)) == fr.dbg_x && int(floorf(ln.image_y)) == fr.dbg_y)
                printf("DEV depth %d prim %d t %.9g o %.9g %.9g %.9g d %.9g %.9g %.9g p %.9g %.9g %.9g nn %.9g %.9g %.9g sn %.9g %.9g %.9g\n", ln.depth, ln.tv.hit_prim, ln.tv.maxt,
                       ln.tv.o.x, ln.tv.o.y, ln.tv.o.z, ln.tv.d.x, ln.tv.d.y, ln.tv.d.z, ln.v.p.x, ln.v.p.y, ln.v.p.z, ln.v.nn.x, ln.v.nn.y, ln.v.nn.z, ln.v.sn.x, ln.v.sn.y, ln.v.sn.z);
#endif
            if (ln.depth == 0) { ln.alpha = 1.f; if (VOL) vol_ray_ptr(fr, 0, gtid)[7 * size_t(fr.n_threads)] = ln.tv.maxt; }   // r.maxt = ray.maxt
            else if (VOL) ln.thr = ln.thr * scene_transmittance<VOL>(sc, ln, ln.tv.o, ln.tv.d, ln.tv.mint, ln.tv.maxt);           // path.cpp:89
            if ((ln.depth == 0 || ln.specular) && ln.v.light >= 0)              // path.cpp:91-92
                ln.L = ln.L + ln.thr * area_L(RT_LIGHT(sc, ln.v.light), vertex_ng<EXT>(ln.v), ln.v.wo);   // isect.Le: dg.nn, the geometric normal
        } else {
            if (!hit) {                                                         // whitted.cpp:52-59
                ln.L = mk3(0.f);
                if (ln.depth == 0) ln.alpha = 0.f;
                ln.stage = ST_RETURN; return;
            }
            make_vertex<EXT>(sc, ln.tv, ln.v);
            if (ln.depth == 0) ln.alpha = 1.f;
            if (VOL) vol_ray_ptr(fr, ln.fsp, gtid)[7 * size_t(fr.n_threads)] = ln.tv.maxt;       // the hit shortens this level's ray
            ln.L = mk3(0.f);
            if (ln.v.light >= 0) ln.L = ln.L + area_L(RT_LIGHT(sc, ln.v.light), vertex_ng<EXT>(ln.v), ln.v.wo);
        }
        ln.li = 0; ln.lj = 0; ln.L_all = mk3(0.f); ln.Ld_light = mk3(0.f);
        ln.stage = ST_DIRECT_NEXT;
        return;
    }
    if constexpr (STAGE == ST_DIRECT_NEXT) {
        const int nLights = int(sc.n_lights);
        if constexpr (WEIGHTED) {
            // WeightedSampleOneLight transport.cpp:71-122, sample offsets of directlighting.cpp:54-64 -- see rt_weighted.h for the three passes
            if (nLights == 0) { ln.stage = ST_SPECULAR; return; }                                   // directlighting.cpp:106
            if (fr.weighted_phase == 1) { ++ln.ord; ln.stage = ST_SPECULAR; return; }             // count: the specular tree alone
            if (fr.weighted_phase == 3 && ln.li > 0) { ln.stage = ST_SPECULAR; return; }
            if (fr.weighted_phase == 2 && fr.wt_mixed) {
                // lights of mixed RNG use: light li with the counter k = lj draws further (every k the frame pass can arrive with, DevFrame::wt_mixed)
                if (ln.li == 0 && ln.lj == 0) { ln.ctr0 = ln.rng.ctr; ln.wt_w = __uint_as_float(1u); }      // wt_w: where the next estimate goes inside the point's record
                if (ln.li >= nLights) { ++ln.ord; ln.rng.ctr = ln.ctr0; ln.lj = 0; ln.stage = ST_SPECULAR; return; }   // the counter stands at k = 0: nothing but the chosen lights' draws moves it
                ln.rng.ctr = ln.ctr0 + uint32_t(ln.lj);
            } else if (fr.weighted_phase == 2) {
                if (ln.li == 0) ln.ctr0 = ln.rng.ctr;
                if (ln.li >= nLights) { ++ln.ord; ln.stage = ST_SPECULAR; return; }                // (the counter stands where one estimate leaves it)
                ln.rng.ctr = ln.ctr0;
            }
            const float ls1 = dim_value(fr, ln, fr.two_d[0], 0, 0), ls2 = dim_value(fr, ln, fr.two_d[0], 0, 1);
            ln.bs1 = dim_value(fr, ln, fr.two_d[1], 0, 0); ln.bs2 = dim_value(fr, ln, fr.two_d[1], 0, 1);
            ln.bcs = dim_value(fr, ln, fr.one_d[1], 0, 0);
            if (fr.weighted_phase == 2) {
                if (ln.li == 0 && ln.lj == 0) RT_GPTR(float, fr.wt_rec)[weighted_record(fr, ln, nLights)] = dim_value(fr, ln, fr.one_d[0], 0, 0);
                estimate_direct_begin<EXT, DEFER>(sc, ln, ln.li, ls1, ls2);
            } else {
                const float2 pick = RT_GPTR(const float2, fr.wt_pick)[ln.ord++];
                ln.li = 1; ln.wt_w = pick.y;
                estimate_direct_begin<EXT, DEFER>(sc, ln, __float_as_int(pick.x), ls1, ls2);
            }
            return;
        }
        if (INTEG == RT_INTEGRATOR_PATH || (INTEG == RT_INTEGRATOR_DIRECT && fr.strategy == RT_STRATEGY_ONE)) {
            // UniformSampleOneLight transport.cpp:51-70
            if (nLights == 0 || ln.li > 0) { ln.stage = (INTEG == RT_INTEGRATOR_PATH) ? ST_BOUNCE : ST_SPECULAR; return; }
            ln.li = 1;
            const int k = (INTEG == RT_INTEGRATOR_PATH) ? ln.depth : 0;
            const bool from_sampler = (INTEG == RT_INTEGRATOR_DIRECT) || k < 3;   // SAMPLE_DEPTH path.cpp:40
            float un, ls1, ls2;
            if (from_sampler) {
                const int i1 = (INTEG == RT_INTEGRATOR_PATH) ? 3 * k : 0;
                const int i2 = (INTEG == RT_INTEGRATOR_PATH) ? 3 * k : 0;
                un = dim_value(fr, ln, fr.one_d[i1], 0, 0);
                ls1 = dim_value(fr, ln, fr.two_d[i2], 0, 0); ls2 = dim_value(fr, ln, fr.two_d[i2], 0, 1);
                ln.bs1 = dim_value(fr, ln, fr.two_d[i2 + 1], 0, 0); ln.bs2 = dim_value(fr, ln, fr.two_d[i2 + 1], 0, 1);
                ln.bcs = dim_value(fr, ln, fr.one_d[i1 + 1], 0, 0);
            } else {
                un = ln.rng.next_float();
                ls1 = ln.rng.next_float(); ls2 = ln.rng.next_float();           // transport.cpp:141-145
                ln.bs1 = ln.rng.next_float(); ln.bs2 = ln.rng.next_float(); ln.bcs = ln.rng.next_float();
            }
            int lightNum = min(int(floorf(un * nLights)), nLights - 1);
            estimate_direct_begin<EXT, DEFER>(sc, ln, lightNum, ls1, ls2);
            return;
        }
        if (INTEG == RT_INTEGRATOR_DIRECT) {
            // UniformSampleAllLights transport.cpp:31-50 with the dimensions of directlighting.cpp:46-53
            if (ln.li >= nLights) { ln.L = ln.L + ln.L_all; ln.stage = ST_SPECULAR; return; }
            const DimReq RT_G *ld = RT_GPTR(const DimReq, fr.light_dims) + 3 * ln.li;
            const DimReq rl = ld[0], rb = ld[1], rc = ld[2];
            if (ln.lj == 0) ln.Ld_light = mk3(0.f);
            const float ls1 = dim_value(fr, ln, rl, ln.lj, 0), ls2 = dim_value(fr, ln, rl, ln.lj, 1);
            ln.bs1 = dim_value(fr, ln, rb, ln.lj, 0); ln.bs2 = dim_value(fr, ln, rb, ln.lj, 1);
            ln.bcs = dim_value(fr, ln, rc, ln.lj, 0);
            estimate_direct_begin<EXT, DEFER>(sc, ln, ln.li, ls1, ls2);
            return;
        }
        // Whitted: one sample per light, unweighted (whitted.cpp:73-81)
        if (ln.li >= nLights) { ln.stage = ST_SPECULAR; return; }
        {
            LightRef Lt = RT_LIGHT(sc, ln.li);
            MatRef m = RT_MAT(sc, ln.v.mat);
            const int cur = ln.li++;
            V3 wi, Li, sd; float smax;
            if (light_is_delta(Lt)) Li = delta_light_sample(Lt, ln.v.p, wi, sd, smax);   // point.cpp:55-60, spot.cpp:61-67, distant.cpp:57-62
            else {                                                              // area.cpp:96-105
                float u2 = ln.rng.next_float();     // g++ evaluates the two RandomFloat() arguments right to left
                float u1 = ln.rng.next_float();
                V3 ns; V3 ps = area_sample_point<EXT>(sc, Lt, ln.v.p, u1, u2, ln.rng, ns);
                wi = normalize3(ps - ln.v.p);
                float pdf = area_light_pdf<EXT>(sc, Lt, ln.v.p, wi);
                Li = (pdf == 0.f) ? mk3(0.f) : div_s(area_L(Lt, ns, -wi), pdf);
                sd = ps - ln.v.p; smax = 1.f - RT_RAY_EPSILON;
            }
            (void)cur;
            if (is_black(Li)) return;
            V3 f = bsdf_f<EXT>(m, ln.v, ln.v.wo, wi);
            if (is_black(f)) return;
            ln.pend = (f * Li) * absdot3(wi, ln.v.nn);
            launch_ray<DEFER>(ln, sc, ln.v.p, sd, RT_RAY_EPSILON, smax, true, ST_SHADOW_DONE);
        }
        return;
    }
    if constexpr (STAGE == ST_SHADOW_DONE) {
        if (COUNT) ++*c_any;
        const bool occluded = ln.tv.hit_prim >= 0;
        if (INTEG == RT_INTEGRATOR_WHITTED) {
            if (!occluded) ln.L = ln.L + ln.pend * scene_transmittance<VOL>(sc, ln, ln.tv.o, ln.tv.d, ln.tv.mint, ln.tv.maxt);   // whitted.cpp:80
            ln.stage = ST_DIRECT_NEXT;
            return;
        }
        if (!occluded) ln.Ld = ln.Ld + ln.pend * scene_transmittance<VOL>(sc, ln, ln.tv.o, ln.tv.d, ln.tv.mint, ln.tv.maxt);       // transport.cpp:155
        ln.stage = ST_ED_BSDF;
        return;
    }
    if constexpr (STAGE == ST_ED_BSDF) {
        estimate_direct_bsdf<EXT, DEFER>(sc, ln);
        return;
    }
    if constexpr (STAGE == ST_MIS_DONE) {
        if (COUNT) ++*c_closest;
        if (ln.tv.hit_prim >= 0) {                                              // transport.cpp:180-184
            V3 nh; int light;
            prim_normal_light<EXT>(sc, ln.tv, nh, light);
            if (light == ln.cur_light) {
                if (dot3(nh, -ln.tv.d) > 0)                                    // isect.Le(-wi) non-black; transport.cpp:188-190
                    ln.Ld = ln.Ld + ln.pend * scene_transmittance<VOL>(sc, ln, ln.tv.o, ln.tv.d, ln.tv.mint, ln.tv.maxt);
            }
        }
        ln.stage = ST_ED_DONE;
        return;
    }
    if constexpr (STAGE == ST_ED_DONE) {
        const int nLights = int(sc.n_lights);
        if constexpr (WEIGHTED) {
            if (fr.weighted_phase == 2) {                                       // survey: what L.y() would be had this light been the chosen one
                const unsigned at = fr.wt_mixed ? __float_as_uint(ln.wt_w) : 1u + 2u * unsigned(ln.li);
                float RT_G *rec = RT_GPTR(float, fr.wt_rec) + weighted_record(fr, ln, nLights) + at;
                rec[0] = lum_y(ln.Ld);                                          // transport.cpp:113 (weighted branch)
                rec[1] = lum_y(ln.Ld * float(nLights));                         // transport.cpp:95  (start-up branch: L = nLights * EstimateDirect)
                if (fr.wt_mixed) {
                    ln.wt_w = __uint_as_float(at + 2u);
                    const unsigned j = ln.ord - RT_GPTR(const unsigned, fr.wt_base)[ln.work];      // this is the sample's j-th shading point: k = 0 .. j
                    if (light_draws_rng(RT_LIGHT(sc, ln.li)) && unsigned(ln.lj) < j) ++ln.lj; else { ln.lj = 0; ++ln.li; }
                } else ++ln.li;
                ln.stage = ST_DIRECT_NEXT;
            } else {
                if (ln.wt_w == 0.f) ln.L = ln.L + ln.Ld * float(nLights);       // UniformSampleOneLight transport.cpp:66-69
                else ln.L = ln.L + div_s(ln.Ld, ln.wt_w);                       // L /= lightSampleWeight transport.cpp:118
                ln.stage = ST_SPECULAR;
            }
            return;
        }
        if (INTEG == RT_INTEGRATOR_PATH) {
            ln.L = ln.L + ln.thr * (ln.Ld * float(nLights));                    // path.cpp:99-110
            ln.stage = ST_BOUNCE;
        } else if (fr.strategy == RT_STRATEGY_ONE) {
            ln.L = ln.L + ln.Ld * float(nLights);
            ln.stage = ST_SPECULAR;
        } else {
            ln.Ld_light = ln.Ld_light + ln.Ld;
            const int ns = RT_GPTR(const DimReq, fr.light_dims)[3 * ln.li].n;
            if (++ln.lj >= ns) {
                ln.L_all = ln.L_all + ln.Ld_light * (1.f / float(ns));           // L += Ld / nSamples
                ln.lj = 0; ++ln.li;
            }
            ln.stage = ST_DIRECT_NEXT;
        }
        return;
    }
    if constexpr (STAGE == ST_BOUNCE) {                                                           // path.cpp:111-143
        MatRef m = RT_MAT(sc, ln.v.mat);
        const int k = ln.depth;
        float bs1, bs2, bcs;
        if (k < 3) {
            bs1 = dim_value(fr, ln, fr.two_d[3 * k + 2], 0, 0); bs2 = dim_value(fr, ln, fr.two_d[3 * k + 2], 0, 1);
            bcs = dim_value(fr, ln, fr.one_d[3 * k + 2], 0, 0);
        } else { bs1 = ln.rng.next_float(); bs2 = ln.rng.next_float(); bcs = ln.rng.next_float(); }
        V3 wi; float pdf; int flags;
        V3 f = bsdf_sample_f<EXT>(m, ln.v, ln.v.wo, wi, bs1, bs2, bcs, pdf, BX_ALL, flags);
        if (is_black(f) || pdf == 0.f) { ln.stage = ST_RETURN; return; }
        ln.specular = (flags & BX_SPECULAR) != 0;
        ln.thr = ln.thr * div_s(f * absdot3(wi, ln.v.nn), pdf);
        if (k > 3) {
            if (ln.rng.next_float() > .5f) { ln.stage = ST_RETURN; return; }
            ln.thr = div_s(ln.thr, .5f);
        }
        if (k == fr.max_depth) { ln.stage = ST_RETURN; return; }
        ++ln.depth;
        launch_ray<DEFER>(ln, sc, ln.v.p, wi, RT_RAY_EPSILON, RT_INF, false, ST_VERTEX);
        return;
    }
    if constexpr (STAGE == ST_SPECULAR) {                                                         // whitted.cpp:82-109
        if (!(ln.depth < fr.max_depth)) { ln.stage = ST_RETURN; return; }       // rayDepth++ < maxDepth
        MatRef m = RT_MAT(sc, ln.v.mat);
        float u3 = ln.rng.next_float(), u2 = ln.rng.next_float(), u1 = ln.rng.next_float();
        V3 wi; float pdf; int flags;
        V3 f = bsdf_sample_f<EXT>(m, ln.v, ln.v.wo, wi, u1, u2, u3, pdf, BX_REFLECTION | BX_SPECULAR, flags);
        if (!is_black(f) && pdf > 0.f) f = div_s(f, pdf);                       // reflection.cpp:399
        const float ad = absdot3(wi, ln.v.nn);
        if (!is_black(f) && ad > 0.f) {
            frame_push<EXT>(fr, ln, gtid, f, ad, ST_SPEC_TRANS);
            ++ln.depth;
            launch_ray<DEFER>(ln, sc, ln.v.p, wi, RT_RAY_EPSILON, RT_INF, false, ST_VERTEX);
            if (VOL) { Ray cr; cr.o = ln.v.p; cr.d = wi; cr.mint = RT_RAY_EPSILON; cr.maxt = RT_INF; vol_store_ray(fr, ln.fsp, gtid, cr); }
            return;
        }
        ln.stage = ST_SPEC_TRANS;
        return;
    }
    if constexpr (STAGE == ST_SPEC_TRANS) {                                                       // whitted.cpp:110-135
        MatRef m = RT_MAT(sc, ln.v.mat);
        float u3 = ln.rng.next_float(), u2 = ln.rng.next_float(), u1 = ln.rng.next_float();
        V3 wi; float pdf; int flags;
        V3 f = bsdf_sample_f<EXT>(m, ln.v, ln.v.wo, wi, u1, u2, u3, pdf, BX_TRANSMISSION | BX_SPECULAR, flags);
        if (!is_black(f) && pdf > 0.f) f = div_s(f, pdf);
        const float ad = absdot3(wi, ln.v.nn);
        if (!is_black(f) && ad > 0.f) {
            frame_push<EXT>(fr, ln, gtid, f, ad, ST_RETURN);
            ++ln.depth;
            launch_ray<DEFER>(ln, sc, ln.v.p, wi, RT_RAY_EPSILON, RT_INF, false, ST_VERTEX);
            if (VOL) { Ray cr; cr.o = ln.v.p; cr.d = wi; cr.mint = RT_RAY_EPSILON; cr.maxt = RT_INF; vol_store_ray(fr, ln.fsp, gtid, cr); }
            return;
        }
        ln.stage = ST_RETURN;
        return;
    }
    if constexpr (STAGE == ST_RETURN) {          // the surface integrator's Li for the current level is complete in ln.L
        ln.stage = VOL ? ST_VOL_BEGIN : ST_POP;
        return;
    }
    if constexpr (VOL && (STAGE == ST_VOL_BEGIN || STAGE == ST_VOL_STEP)) {
        // Scene::Li = T * Lo + Lv (scene.cpp:120-126): the volume integrator along this level's ray.  The march's state is parked in HBM scratch
        // (vol_state[13][thread or slot]) while a step's shadow ray is traced (megakernel) or until rt::pipe_march_kernel picks the slot up
        // (queue pipeline, PARK: only the head of the march runs here, the slot stays in ST_VOL_STEP without a ray = "parked")
        const Ray ray = vol_load_ray(fr, ln.fsp, gtid);
        const size_t st = fr.n_threads;
        float RT_G *vs = RT_GPTR(float, fr.vol_state) + gtid;
        float RT_G *samp = RT_GPTR(float, fr.vol_samp) + gtid;
        March m;
        bool park;
        if (STAGE == ST_VOL_BEGIN) {
            march_begin(sc, fr, ln, ray, m, samp, st);
            if (PARK) { ln.stage = ST_VOL_STEP; park = true; }
            else park = march_steps<COUNT, EXT, DEFER>(sc, fr, ln, ray, m, samp, st, false, c_any);
        } else {                                                                   // resume after the step's shadow ray
            m.i = __float_as_int(vs[0]); m.N = __float_as_int(vs[st]); m.t0 = vs[2 * st]; m.step = vs[3 * st];
            m.Tr = mk3(vs[4 * st], vs[5 * st], vs[6 * st]); m.p = mk3(vs[7 * st], vs[8 * st], vs[9 * st]); m.Lv = mk3(vs[10 * st], vs[11 * st], vs[12 * st]);
            park = march_steps<COUNT, EXT, DEFER>(sc, fr, ln, ray, m, samp, st, true, c_any);
        }
        if (park) {
            vs[0] = __int_as_float(m.i); vs[st] = __int_as_float(m.N); vs[2 * st] = m.t0; vs[3 * st] = m.step;
            vs[4 * st] = m.Tr.x; vs[5 * st] = m.Tr.y; vs[6 * st] = m.Tr.z; vs[7 * st] = m.p.x; vs[8 * st] = m.p.y; vs[9 * st] = m.p.z;
            vs[10 * st] = m.Lv.x; vs[11 * st] = m.Lv.y; vs[12 * st] = m.Lv.z;
        }
        return;
    }
    if constexpr (STAGE == ST_POP) {
        if (ln.fsp == 0) { ln.stage = ST_FINISH; return; }
        ln.stage = frame_pop<EXT>(fr, ln, gtid, ln.L);
        return;
    }
    if constexpr (STAGE == ST_FINISH) {
        if (WEIGHTED && fr.weighted_phase == 1) RT_GPTR(unsigned, fr.wt_base)[ln.work] = ln.ord;       // (ord started at 0: the sample's shading points)
        if (!WEIGHTED || fr.weighted_phase == 3) sample_write(fr, ln, ln.L, ln.alpha, *c_bad);
        ln.stage = ST_FETCH;
        return;
    }
    (void)gtid; (void)c_closest; (void)c_any; (void)c_bad;
}

// ---- the path integrator BY VERTEX in the megakernel (round 5) -----------------------------------------------------------------------------
// Nothing PathIntegrator::Li does at a vertex depends on what the vertex's rays return (without a medium): the sample values and every RandomFloat() are
// consumed in an order no ray result changes, the continuation direction is sampled from the BSDF alone, and the two halves of EstimateDirect only
// decide whether two already computed contributions count (rt_pipe_vertex.h runs the queue pipeline on that).  Per ray, a lane needs a shading pass
// after EVERY ray -- VERTEX + DIRECT_NEXT, ED_BSDF, ED_DONE + BOUNCE -- and each pass finds a third of the finished lanes in each of those stages: the
// stage bodies run at a third of the occupancy they could have.  Here one pass does the whole vertex (the reference's operations in the reference's
// order: stage_body, DEFER form), the shadow ray starts at once, the MIS ray and the continuation ray wait in registers and are started INSIDE the
// traversal loop when the lane's previous ray ends (byv_ray_advance: ~50 instructions, no shading pass), and the two EstimateDirect terms are added
//     Ld = 0; if (unoccluded) Ld += pendS; if (the MIS ray hit the sampled emitter's front) Ld += pendM; L += thr_old * (Ld * nLights)
// -- the reference's sums in the reference's order (transport.cpp:127,155,190; path.cpp:99-110) -- when the lane comes back for its next vertex.
// Same rays, same arithmetic: films bit-identical to the per-ray form (the counting twins keep it; tests compare).
enum { BV_S = 1u, BV_M = 2u, BV_B = 4u, BV_ED = 8u,            // the vertex has a shadow / MIS / continuation ray; EstimateDirect's sum is pending
       BV_QM = 16u, BV_QB = 32u,                                // MIS / continuation ray not started yet
       BV_CUR_S = 64u, BV_CUR_M = 128u, BV_CUR_B = 256u,        // the ray in flight
       BV_S_CLEAR = 512u, BV_M_LIT = 1024u };                   // results: the shadow ray was unoccluded; the MIS ray hit the sampled emitter from its front
// what the ray that has just ended says (transport.cpp:152-156, :180-190)
template <bool COUNT, bool EXT>
RT_DEV void byv_ray_result(const DevScene &sc, Lane &ln, unsigned *c_closest, unsigned *c_any) {
    if (ln.vf & BV_CUR_S) {
        if (COUNT) ++*c_any;
        if (ln.tv.hit_prim < 0) ln.vf |= BV_S_CLEAR;
    } else if (ln.vf & BV_CUR_M) {
        if (COUNT) ++*c_closest;
        if (ln.tv.hit_prim >= 0) {
            V3 nh; int light;
            prim_normal_light<EXT>(sc, ln.tv, nh, light);
            if (light == ln.cur_light && dot3(nh, -ln.tv.d) > 0) ln.vf |= BV_M_LIT;
        }
    }
    ln.vf &= ~(BV_CUR_S | BV_CUR_M | BV_CUR_B);
}
// inside the traversal loop: the lane's ray has ended and its vertex has another one waiting
template <bool COUNT, int ACCEL, bool EXT>
RT_DEV void byv_ray_advance(const DevScene &sc, Lane &ln, unsigned *c_closest, unsigned *c_any) {
    byv_ray_result<COUNT, EXT>(sc, ln, c_closest, c_any);
    Ray r; r.o = ln.tv.o; r.mint = RT_RAY_EPSILON; r.maxt = RT_INF;
    if (ln.vf & BV_QM) { r.d = ln.dM; ln.vf = (ln.vf & ~BV_QM) | BV_CUR_M; }
    else { r.d = ln.dB; ln.vf = (ln.vf & ~BV_QB) | BV_CUR_B; }
    if (ACCEL == RT_ACCEL_GRID) grid_begin(ln.tv, sc, r, false); else trav_begin(ln.tv, sc, r, false);
}
// one pass: every lane without a ray finishes its previous vertex and runs its next one up to its rays (or ends the path: ST_FETCH)
template <bool COUNT, int ACCEL, bool EXT>
RT_DEV void advance_pass_byv(const DevScene &sc, const DevFrame &fr, Lane &ln, unsigned gtid, unsigned *c_closest, unsigned *c_any, unsigned *c_bad) {
    constexpr int INTEG = RT_INTEGRATOR_PATH;
#define RT_BV_BODY(S) stage_body<COUNT, INTEG, false, EXT, S, true>(sc, fr, ln, gtid, c_closest, c_any, c_bad)
    const int nLights = int(sc.n_lights);
    if (!ln.has_ray && (ln.stage == ST_VERTEX || ln.stage == ST_ED_DONE)) {
        byv_ray_result<COUNT, EXT>(sc, ln, c_closest, c_any);          // the vertex's last ray (the earlier ones were read when the next ray started)
        if (ln.vf & BV_ED) {                                            // the two halves of EstimateDirect, then path.cpp:99-110
            V3 Ld = mk3(0.f);                                           // transport.cpp:127
            if (ln.vf & BV_S_CLEAR) Ld = Ld + ln.pendS * mk3(1.f);      // Transmittance = 1 (no medium)
            if (ln.vf & BV_M_LIT) Ld = Ld + ln.pendM * mk3(1.f);
            ln.L = ln.L + ln.thr_old * (Ld * float(nLights));
        }
        ln.vf = 0u;
        if (ln.stage == ST_ED_DONE) ln.stage = ST_RETURN;               // the path ended at that vertex
    }
    unsigned flags = 0;
    V3 o = mk3(0.f), dS = mk3(0.f); float maxtS = 0.f;
    if (!ln.has_ray && ln.stage == ST_VERTEX) RT_BV_BODY(ST_VERTEX);    // -> ST_DIRECT_NEXT, or ST_RETURN (the ray left the scene)
    if (!ln.has_ray && ln.stage == ST_DIRECT_NEXT) {                    // UniformSampleOneLight -> EstimateDirect (transport.cpp:51-70, 123-194)
        RT_BV_BODY(ST_DIRECT_NEXT);                                     // light-sampling half: shadow ray recorded in ln.tv, or ST_ED_BSDF, or (no lights) ST_BOUNCE
        if (ln.has_ray) { ln.has_ray = false; ln.stage = ST_ED_BSDF; flags |= BV_S; ln.pendS = ln.pend; o = ln.tv.o; dS = ln.tv.d; maxtS = ln.tv.maxt; }
        if (ln.stage == ST_ED_BSDF) {
            RT_BV_BODY(ST_ED_BSDF);                                     // BSDF-sampling half: MIS ray recorded, or ST_ED_DONE
            if (ln.has_ray) { ln.has_ray = false; flags |= BV_M | BV_QM; ln.pendM = ln.pend; o = ln.tv.o; ln.dM = ln.tv.d; }
            if (flags & (BV_S | BV_M)) { flags |= BV_ED; ln.thr_old = ln.thr; }     // L += thr * (Ld * nLights) waits for the rays
            else { ln.stage = ST_ED_DONE; RT_BV_BODY(ST_ED_DONE); }                  // Ld = 0: nothing to wait for
            ln.stage = ST_BOUNCE;
        }
    }
    if (!ln.has_ray && ln.stage == ST_BOUNCE) {                         // path.cpp:111-143
        RT_BV_BODY(ST_BOUNCE);
        if (ln.has_ray) { ln.has_ray = false; flags |= BV_B | BV_QB; o = ln.tv.o; ln.dB = ln.tv.d; }
    }
    if (!ln.has_ray && ln.stage == ST_RETURN && !(flags & BV_ED)) { RT_BV_BODY(ST_RETURN); RT_BV_BODY(ST_POP); RT_BV_BODY(ST_FINISH); }      // -> ST_FETCH
    if (flags & (BV_S | BV_M | BV_B)) {                                 // start the vertex's first ray
        Ray r; r.o = o; r.mint = RT_RAY_EPSILON;
        bool any = false;
        if (flags & BV_S) { r.d = dS; r.maxt = maxtS; any = true; flags |= BV_CUR_S; }
        else if (flags & BV_M) { r.d = ln.dM; r.maxt = RT_INF; flags = (flags & ~BV_QM) | BV_CUR_M; }
        else { r.d = ln.dB; r.maxt = RT_INF; flags = (flags & ~BV_QB) | BV_CUR_B; }
        if (ACCEL == RT_ACCEL_GRID) grid_begin(ln.tv, sc, r, any); else trav_begin(ln.tv, sc, r, any);
        ln.vf = flags;
        ln.has_ray = true;
        ln.stage = (flags & BV_B) ? ST_VERTEX : ST_ED_DONE;            // where the lane resumes when the last of these rays is back
    }
#undef RT_BV_BODY
}

// (The same for DirectLightingIntegrator::Li -- both rays of one EstimateDirect per pass, the vertex kept live for the next light and the specular recursion -- was
// built, is bit-identical, and is 6 % SLOWER on C3 (kernel 12.07 -> 12.79 ms: the kernel sits at the 128-register cap with nothing dying across the rays, 11 spilled
// registers); profiles/r05_direct_by_estimate_experiment.patch, profiles/r05_by_vertex_scan.txt.)

// One shading pass.  The stages are visited in pipeline order, so a lane flows through as many of them as it can
// in a single pass, and -- the point of the ordering -- lanes that entered at different stages *merge*: e.g. the
// bounce-sampling block runs once per pass for every lane that needs it, whether the lane just came back from a
// shadow ray, from a MIS ray or straight from a vertex.  (A switch executed once per transition would run every
// stage body once per lane phase: ~8x lower SIMD utilisation with 64 lanes at random phases.)  Backward edges
// (next light of the all-lights loop, popping a recursion frame) simply take another pass.
// Phase gating (path integrator): a path alternates [closest-hit ray -> VERTEX, DIRECT_NEXT -> shadow ray] and
// [shadow ray -> SHADOW_DONE .. BOUNCE -> closest-hit ray]; 64 lanes at random phases run every pass of a sweep half
// empty.  With `phase` 0 / 1 a sweep runs only the first / second group and the kernel alternates them, so the lanes of
// a wave fall into step (a lane that leaves the rhythm -- its ray missed, its light sample was black -- idles one
// trace and is back in step).  phase < 0: no gating.  The per-lane order of stages is unchanged.
RT_DEV bool stage_in_phase(int stage, int phase) {
    if (stage == ST_EXIT) return false;
    if (phase < 0) return true;
    const bool first = stage == ST_VERTEX || stage == ST_DIRECT_NEXT;
    return phase == 0 ? first : !first;
}
// PARK (queue pipeline with a medium): only the head of a ray march is run here (march_begin) -- the lane then sits in ST_VOL_STEP without a ray
// and its steps are run by rt::pipe_march_kernel (rt_pipe_march.h), which hands it back in ST_POP.
template <bool COUNT, int INTEG_, bool VOL, bool EXT, bool DEFER = false, bool PARK = false>
RT_DEV void advance_pass(const DevScene &sc, const DevFrame &fr, Lane &ln, unsigned gtid,
                         unsigned *c_closest, unsigned *c_any, unsigned *c_bad, int phase) {
    constexpr int INTEG = INTEG_ == RT_INTEG_DIRECT_WEIGHTED ? int(RT_INTEGRATOR_DIRECT) : INTEG_;
#ifdef RT_PROFILE_STAGES
#define RT_RUN(S) { const unsigned long long m_ = __ballot(!ln.has_ray && ln.stage == S); if (m_) { const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (!ln.has_ray && ln.stage == S) stage_body<COUNT, INTEG_, VOL, EXT, S, DEFER, PARK>(sc, fr, ln, gtid, c_closest, c_any, c_bad); \
        if (__lane_id() == 0) { atomicAdd(fr.counters + 24 + 2 * S, __builtin_readcyclecounter() - t_); atomicAdd(fr.counters + 24 + 2 * S + 1, (unsigned long long)__popcll(m_) | (1ull << 40)); } } }
#else
#define RT_RUN(S) if (!ln.has_ray && ln.stage == S) stage_body<COUNT, INTEG_, VOL, EXT, S, DEFER, PARK>(sc, fr, ln, gtid, c_closest, c_any, c_bad)
#endif
    if (phase != 0) {
        RT_RUN(ST_MIS_DONE);
        RT_RUN(ST_SHADOW_DONE);
    }
    if (phase != 1) {
        RT_RUN(ST_VERTEX);
        RT_RUN(ST_DIRECT_NEXT);
    }
    if (phase != 0) {
        if (INTEG != RT_INTEGRATOR_WHITTED) { RT_RUN(ST_ED_BSDF); RT_RUN(ST_ED_DONE); }
        if (INTEG == RT_INTEGRATOR_PATH) { RT_RUN(ST_BOUNCE); }
        else { RT_RUN(ST_SPECULAR); RT_RUN(ST_SPEC_TRANS); }
        RT_RUN(ST_RETURN);
        if (VOL) { if (!PARK) RT_RUN(ST_VOL_STEP); RT_RUN(ST_VOL_BEGIN); }
        RT_RUN(ST_POP);
        RT_RUN(ST_FINISH);
    }
#undef RT_RUN
}

}  // namespace rt
