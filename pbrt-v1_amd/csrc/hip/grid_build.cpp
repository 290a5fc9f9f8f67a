// grid_build.cpp -- host-side uniform-grid construction (flat, eager) for the HIP traversal kernels.
//
// Behavioural contract = GridAccel's constructor over fully refined primitives, i.e. the reference's
// "refineimmediately" form (accelerators/grid.cpp:122-210): bounds = union of primitive bounds; voxels per axis
// = clamp(Round2Int(extent * 3*cbrt(N) / maxExtent), 1, 64) (:141-152); Width/InvWidth (:154-158); a primitive is
// added, in primitive order, to every voxel its bounding box overlaps, PosToVoxel truncating then clamping (:95-99,
// :163-183).  The reference's default is a lazily refined two-level grid (whole meshes in the top grid, one nested
// grid per mesh built at first hit, :292-310); closest-hit / any-hit RESULTS do not depend on that structure, so the
// GPU gets the eager flat form -- only the work counters differ from the lazy form.
// Voxel encoding: uint2 {offset into the reference list, primitive count} (count 0 = empty voxel).
#include "rt_internal.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <thread>

namespace rt {

static inline int round_to_int(double v) { return int(v + (.5 - 1.4e-11)); }     // pbrt.h:604-613

void build_grid(const float *tri_verts, uint32_t n_tris, GridAccelData &out) {
    auto t0 = std::chrono::steady_clock::now();
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    std::vector<float> blo(size_t(3) * n_tris), bhi(size_t(3) * n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) {
        const float *v = tri_verts + size_t(9) * i;
        for (int a = 0; a < 3; ++a) {
            float mn = std::min(std::min(v[a], v[3 + a]), v[6 + a]), mx = std::max(std::max(v[a], v[3 + a]), v[6 + a]);
            blo[3 * i + a] = mn; bhi[3 * i + a] = mx;
            lo[a] = std::min(lo[a], mn); hi[a] = std::max(hi[a], mx);
        }
    }
    for (int a = 0; a < 3; ++a) { out.bounds[a] = lo[a]; out.bounds[3 + a] = hi[a]; }
    if (n_tris == 0) {                                   // empty world: a single empty voxel
        for (int a = 0; a < 3; ++a) { out.nvox[a] = 1; out.width[a] = 0.f; out.inv_width[a] = 0.f; }
        out.voxels.assign(1, Node{0, 0}); out.refs.clear(); return;
    }
    const float delta[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    const int maxAxis = (delta[0] > delta[1] && delta[0] > delta[2]) ? 0 : ((delta[1] > delta[2]) ? 1 : 2);   // geometry.h:265-273
    const float invMaxWidth = 1.f / delta[maxAxis];
    const float cubeRoot = 3.f * powf(float(n_tris), 1.f / 3.f);
    const float voxelsPerUnitDist = cubeRoot * invMaxWidth;
    for (int a = 0; a < 3; ++a) {
        int nv = round_to_int(delta[a] * voxelsPerUnitDist);
        out.nvox[a] = nv < 1 ? 1 : (nv > 64 ? 64 : nv);
        out.width[a] = delta[a] / out.nvox[a];
        out.inv_width[a] = (out.width[a] == 0.f) ? 0.f : 1.f / out.width[a];
    }
    auto pos_to_voxel = [&](float p, int a) {
        int v = int((p - lo[a]) * out.inv_width[a]);                           // Float2Int = truncation
        return v < 0 ? 0 : (v > out.nvox[a] - 1 ? out.nvox[a] - 1 : v);
    };
    const size_t nv = size_t(out.nvox[0]) * out.nvox[1] * out.nvox[2];
    out.voxels.assign(nv, Node{0, 0});
    // pass 1: counts, pass 2: fill in primitive order (== AddPrimitive order)
    std::vector<int> ext(size_t(6) * n_tris);
    for (uint32_t i = 0; i < n_tris; ++i)
        for (int a = 0; a < 3; ++a) { ext[6 * i + a] = pos_to_voxel(blo[3 * i + a], a); ext[6 * i + 3 + a] = pos_to_voxel(bhi[3 * i + a], a); }
    // Voxel lists must hold their primitives in primitive order (== the reference's AddPrimitive order).  Parallel form: the grid is
    // cut into z-slabs, one per thread; every thread walks ALL primitives in order but touches only the voxels of its own slab, so
    // no two threads write the same voxel and every list comes out in primitive order, exactly as in the serial loop.
    int threads = int(std::thread::hardware_concurrency());
    threads = std::max(1, std::min(std::min(threads, 32), out.nvox[2]));
    if (n_tris < 50000) threads = 1;
    auto slab_pass = [&](auto &&fn) {
        auto work = [&](int t) {
            const int z_lo = int((long long)out.nvox[2] * t / threads), z_hi = int((long long)out.nvox[2] * (t + 1) / threads) - 1;
            for (uint32_t i = 0; i < n_tris; ++i) {
                const int za = std::max(ext[6 * i + 2], z_lo), zb = std::min(ext[6 * i + 5], z_hi);
                for (int z = za; z <= zb; ++z)
                    for (int y = ext[6 * i + 1]; y <= ext[6 * i + 4]; ++y)
                        for (int x = ext[6 * i]; x <= ext[6 * i + 3]; ++x)
                            fn(size_t(z) * out.nvox[0] * out.nvox[1] + size_t(y) * out.nvox[0] + x, i);      // Offset() grid.cpp:106-108
            }
        };
        if (threads == 1) { work(0); return; }
        ThreadGroup pool; for (int t = 0; t < threads; ++t) pool.spawn(work, t); pool.join();
    };
    slab_pass([&](size_t o, uint32_t) { ++out.voxels[o].y; });
    uint32_t total = 0;
    for (size_t o = 0; o < nv; ++o) { out.voxels[o].x = total; total += out.voxels[o].y; out.voxels[o].y = 0; }
    out.refs.resize(total);
    slab_pass([&](size_t o, uint32_t i) { out.refs[out.voxels[o].x + out.voxels[o].y++] = i; });
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace rt
