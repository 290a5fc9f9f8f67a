// rt_scene.hip -- rt_scene_create: the flattened accelerator (kd_build.cpp / grid_build.cpp, or a prebuilt one) and everything the kernels read derived from it and
// uploaded -- primitive records and leaf entries, sibling-pair blocks, shading constants, lights -- plus the launch geometry of the persistent kernels;
// rt_scene_destroy, and the accelerator-only entry points (rt_accel_* / rt_kdtree_*) of the C ABI.
#include "rt_host.h"

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

static void fill_info(const KdTree &tree, const GridAccelData &g, int kind, uint32_t n_tris, RtAccelInfo *info);
extern "C" int rt_scene_destroy(RtScene *s);

extern "C" {

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
// tree (pbrt_hip.h RtAccelInfo / rt_accel_copy); everything the device derives from them (primitive records and leaf entries, pair blocks) is rebuilt here.
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
    // ... and so do the leaf entries (four passes over the leaf lists by all host threads: 1.3 s at 10 M triangles) while this thread fills the per-triangle records
    // and uploads the tree.  Scenes of a few thousand references (C2's 14 triangles: cache resident, bound by instruction issue -- the entry form costs it 2.8 %,
    // profiles/r06_dedup_scan.txt) get runs of consecutive records per leaf; everything larger shares one record per primitive.
    LeafLayout ll;
    int ll_status = 0;                                  // 1 laid out, -1 index ranges exceeded, -2 out of memory
    bool runs = s->tree.leaf_refs.size() + s->tree.nodes.size() / 2 <= 32768;
    if (const char *e = knob("PBRT_HIP_LEAF_RUNS")) runs = std::atoi(e) != 0;
    const bool copies = knob("PBRT_HIP_LEAF_COPIES") != nullptr;
    std::thread layout_thread;
    Joiner joiner2{layout_thread};
    if (s->accel_kind == RT_ACCEL_KDTREE) layout_thread = std::thread([&] {
        try { ll_status = leaf_cursor_layout(s->tree.nodes, s->tree.leaf_refs, d->n_tris, copies, runs, ll) ? 1 : -1; } catch (...) { ll_status = -2; }
    });
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
        tick("node / leaf-list upload");
        layout_thread.join();
        if (ll_status == -2) return fail(RT_ENOMEM, "rt_scene_create: out of host memory while laying out the leaf entries");
        if (ll_status != 1) return fail(RT_EINVAL, "rt_scene_create: primitive records beyond 2^30 float4 units or leaf entries beyond 2^31");
        s->dev.leaf_runs = runs ? 1u : 0u;
        const NodeVec &tn = ll.tnodes;
        tick("leaf entries (rest)");
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
// The leaf layout rt_scene_create derives from a kd-tree (leaf_layout.cpp), on the host alone: what the CPU tests check the entry encoding with.
int rt_accel_leaf_layout(const RtAccel *t, int runs, int copies, uint32_t *tnodes, uint32_t *slot_prim, uint32_t *entries, RtLeafLayoutInfo *info) {
    if (!t || !info) return fail(RT_EINVAL, "rt_accel_leaf_layout: null argument");
    if (t->kind != RT_ACCEL_KDTREE) return fail(RT_EINVAL, "rt_accel_leaf_layout: not a kd-tree");
    return guarded("rt_accel_leaf_layout", [&] {
        LeafLayout ll;
        if (!leaf_cursor_layout(t->tree.nodes, t->tree.leaf_refs, t->n_tris, copies != 0, runs != 0, ll)) return fail(RT_EINVAL, "rt_accel_leaf_layout: primitive records beyond 2^30 float4 units or leaf entries beyond 2^31");
        info->n_nodes = ll.tnodes.size(); info->n_slots = ll.n_slots; info->n_entries = ll.lrefs.size(); info->stride = RT_TRI_STRIDE;
        if (tnodes && !ll.tnodes.empty()) std::memcpy(tnodes, ll.tnodes.data(), ll.tnodes.size() * sizeof(Node));
        if (slot_prim && ll.n_slots) std::memcpy(slot_prim, ll.slot_prim.data(), ll.n_slots * sizeof(uint32_t));
        if (entries) std::memcpy(entries, ll.lrefs.data(), ll.lrefs.size() * sizeof(uint32_t));
        return int(RT_OK);
    });
}
int rt_kdtree_build(const float *tri_verts, uint32_t n_tris, const RtAccelParams *params, RtKdTree **out) {
    if (params && params->kind != RT_ACCEL_KDTREE) return fail(RT_EINVAL, "rt_kdtree_build: not a kd-tree description");
    return rt_accel_build(tri_verts, n_tris, params, out);
}
int rt_kdtree_info(const RtKdTree *t, RtAccelInfo *info) { return rt_accel_info(t, info); }
int rt_kdtree_copy(const RtKdTree *t, uint32_t *nodes, uint32_t *leaf_refs) { return rt_accel_copy(t, nodes, leaf_refs); }
int rt_kdtree_destroy(RtKdTree *t) { return rt_accel_destroy(t); }

}  // extern "C"
