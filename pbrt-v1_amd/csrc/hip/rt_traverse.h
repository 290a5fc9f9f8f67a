// rt_traverse.h -- kd-tree traversal + ray/triangle test as a resumable per-lane state machine.
//
// Semantics follow KdTreeAccel::Intersect / IntersectP (reference accelerators/kdtree.cpp:313-488),
// BBox::IntersectP (core/geometry.cpp:51-68) and Triangle::Intersect(P)
// (shapes/trianglemesh.cpp:213-314); see SURVEY.md Appendix B items B6, B10-B12.
// MI355X design points:
//   * nodes are 8 B (one global_load_dwordx2 per visit), triangles 48 B (three dwordx4);
//   * the todo stack lives in LDS as stack[entry][lane] (conflict-free ds_write_b64/ds_read_b64),
//     8 B per entry {far child, tmax}: the popped tmin is always the tmax of the leaf just
//     finished, because the leaves visited partition [tmin,tmax] contiguously, so it need not
//     be stored (the reference stores 12 B + pointer);
//   * the LDS ring keeps the top RT_STACK_LDS entries; older ones spill to HBM (counted), so depth is unbounded like MAX_TODO=64
//     never is in practice (<= 17 at 1M triangles);
//   * one step() = one node visit, so a wave can leave the loop when few lanes are still
//     traversing and let the others shade / fetch new rays (persistent-thread scheme).
//   * no mailboxing: re-testing a triangle returns the same t (ties accepted, B6).
#pragma once
#include "rt_device.h"

// LDS stack entries per lane: 12 x 8 B x 256 lanes = 24 KB per workgroup, so LDS never limits residency below 6 workgroups
// per CU; when the ring is full the oldest entry spills to HBM (counted) -- see stack_push / stack_pop.
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 12
#endif
#ifndef RT_BLOCK
#define RT_BLOCK 256
#endif
#include "rt_leaf_entries.h"
// Per-lane 4-entry mailbox window: measured SLOWER on MI355X (C2 92.9 vs 87.8 ms, 100k-soup path 169 vs 165 ms): the
// four compares + rotates per candidate cost more VALU than the avoided re-tests save.  Kept as a compile-time knob.

namespace rt {

// A value the compiler must treat as defined without spending an instruction on it: the select-style steps read their load results on
// every lane and commit them under lane masks, so a lane that did not load needs no particular value -- initialising the eight words of a
// step's two pair records cost 14 v_mov_b32 per step (of ~100 vector instructions).
RT_DEV void undef_u4(uint4 &v) { asm("" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w)); }
RT_DEV void undef_f4(float4 &v) { asm("" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w)); }

struct Ray { V3 o, d; float mint, maxt; };

struct Trav {
    // ray being traced
    V3 o, d, inv;
    float mint, maxt;
    // traversal cursor
    unsigned node;
    unsigned cx, cy;               // sibling-pair form (kdp_step): the CONTENTS of the current node instead of its index
    float tmin, tmax;
    int sp, sbase;                 // todo stack: entries [sbase, sp) live in the LDS ring, [0, sbase) in HBM
    // leaf / voxel primitive-list cursor for the lock-step ("while-while") traversal: at_leaf => test prims [li, ln) of the list at ly;
    // in the flat traversal (one record per primitive, see RT_LE_*): ly = the entry of the primitive to test next, ln_ = the cursor, li unused
    unsigned li, ln_, ly;
    bool at_leaf;
    // grid 3D-DDA cursor (grid.cpp:238-260): voxel position and the ray parameter of the next crossing per axis
    int gpos[3];
    float gnext[3];
    // result
    int hit_prim;       // closest: primitive index or -1; any: 0/-1
    float b1, b2;
    bool any;           // IntersectP semantics
    bool active;
};

struct TravCounters {
    unsigned nodes, leaf_refs, tris, spills;
#ifdef RT_PROFILE
    unsigned long long c_desc, c_leaf, n_chunks, n_pooled, n_iter;
#endif
};
#ifdef RT_PROFILE
#define RT_PFT(x) x
#else
#define RT_PFT(x)
#endif

// Triangle::Intersect up to the acceptance test (trianglemesh.cpp:213-246); exact comparison forms.  The record holds p1 and
// the two edges e1 = p2 - p1, e2 = p3 - p1, subtracted once on the host (the same float operation, so bit-identical).
RT_DEV bool tri_test(V3 p1, V3 e1, V3 e2, V3 o, V3 d, float mint, float maxt, float &t_out, float &b1_out, float &b2_out) {
    V3 s1 = cross3(d, e2);
    float divisor = dot3(s1, e1);
    if (divisor == 0.f) return false;
    float invDivisor = 1.f / divisor;
    V3 dd = o - p1;
    float b1 = dot3(dd, s1) * invDivisor;
    if (b1 < 0.f || b1 > 1.f) return false;
    V3 s2 = cross3(dd, e1);
    float b2 = dot3(d, s2) * invDivisor;
    if (b2 < 0.f || b1 + b2 > 1.f) return false;
    float t = dot3(e2, s2) * invDivisor;
    if (t < mint || t > maxt) return false;
    t_out = t; b1_out = b1; b2_out = b2;
    return true;
}

// ---- quadrics: {Sphere,Disk,Cylinder}::Intersect / IntersectP up to the accepted hit parameter (shapes/sphere.cpp:104-140,
// disk.cpp:64-84, cylinder.cpp:65-107) ----
RT_DEV bool quadratic(float A, float B, float C, float &t0, float &t1) {   // pbrt.h:645-659
    const float discrim = B * B - 4.f * A * C;
    if (discrim < 0.f) return false;
    const float rootDiscrim = sqrtf(discrim);
    float q;
    if (B < 0) q = -.5f * (B - rootDiscrim);
    else q = -.5f * (B + rootDiscrim);
    t0 = q / A;
    t1 = C / q;
    if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
    return true;
}
// phi of an object-space hit point: atan2f(y, x) folded to [0, 2 pi); the hyperboloid measures it against the generating
// segment's point at the hit's height (hyperboloid.cpp:105-111)
RT_DEV float quadric_phi(const DevQuadric RT_G &q, V3 phit) {
    float phi;
    if (q.type == RT_QUADRIC_HYPERBOLOID) {
        const float v = (phit.z - q.p1[2]) / (q.p2[2] - q.p1[2]);
        const V3 pr = mk3(q.p1[0], q.p1[1], q.p1[2]) * (1.f - v) + mk3(q.p2[0], q.p2[1], q.p2[2]) * v;
        phi = atan2f(pr.x * phit.y - phit.x * pr.y, phit.x * pr.x + phit.y * pr.y);
    } else phi = atan2f(phit.y, phit.x);
    if (phi < 0.f) phi += 2.f * RT_PI;
    return phi;
}
RT_DEV bool quadric_test(const DevScene &sc, unsigned qi, V3 ow, V3 dw, float mint, float maxt, float &t_out) {
    const DevQuadric RT_G &q = RT_GPTR(const DevQuadric, sc.quadrics)[qi];
    const V3 o = xform_point(q.w2o, ow), d = xform_vector(q.w2o, dw);           // WorldToObject(r, &ray) transform.h:128-135
    const float radius = q.radius, zmin = q.zmin, zmax = q.zmax, phiMax = q.phi_max;
    if (q.type == RT_QUADRIC_DISK) {                                            // Disk::Intersect(P) disk.cpp:64-123: zmin = height, zmax = innerRadius
        if (double(fabsf(d.z)) < 1e-7) return false;
        const float thit = (zmin - o.z) / d.z;
        if (thit < mint || thit > maxt) return false;
        const V3 phit = o + d * thit;
        const float dist2 = phit.x * phit.x + phit.y * phit.y;
        if (dist2 > radius * radius || dist2 < zmax * zmax) return false;
        float phi = atan2f(phit.y, phit.x);
        if (phi < 0) phi = float(double(phi) + 2. * double(RT_PI));             // "phi += 2. * M_PI": a double sum
        if (phi > phiMax) return false;
        t_out = thit;
        return true;
    }
    // the quadratic forms: sphere.cpp:111-116, cylinder.cpp:72-75, cone.cpp:61-68 (zmax = height), paraboloid.cpp:65-70,
    // hyperboloid.cpp:83-91; then one control flow (the sphere's clip test differs, sphere.cpp:128-130)
    float A, B, C;
    if (q.type == RT_QUADRIC_CONE) {
        float k = radius / zmax;
        k = k * k;
        A = d.x * d.x + d.y * d.y - k * d.z * d.z;
        B = 2 * (d.x * o.x + d.y * o.y - k * d.z * (o.z - zmax));
        C = o.x * o.x + o.y * o.y - k * (o.z - zmax) * (o.z - zmax);
    } else if (q.type == RT_QUADRIC_PARABOLOID) {
        const float k = zmax / (radius * radius);
        A = k * (d.x * d.x + d.y * d.y);
        B = 2 * k * (d.x * o.x + d.y * o.y) - d.z;
        C = k * (o.x * o.x + o.y * o.y) - o.z;
    } else if (q.type == RT_QUADRIC_HYPERBOLOID) {
        const float a = q.a, c = q.c;
        A = a * d.x * d.x + a * d.y * d.y - c * d.z * d.z;
        B = 2.f * (a * d.x * o.x + a * d.y * o.y - c * d.z * o.z);
        C = a * o.x * o.x + a * o.y * o.y - c * o.z * o.z - 1;
    } else if (q.type == RT_QUADRIC_CYLINDER) {
        A = d.x * d.x + d.y * d.y;
        B = 2 * (d.x * o.x + d.y * o.y);
        C = o.x * o.x + o.y * o.y - radius * radius;
    } else {
        A = d.x * d.x + d.y * d.y + d.z * d.z;
        B = 2 * (d.x * o.x + d.y * o.y + d.z * o.z);
        C = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
    }
    float t0, t1;
    if (!quadratic(A, B, C, t0, t1)) return false;
    if (t0 > maxt || t1 < mint) return false;
    float thit = t0;
    if (t0 < mint) { thit = t1; if (thit > maxt) return false; }
    const bool sph = q.type == RT_QUADRIC_SPHERE;
    V3 phit = o + d * thit;
    float phi = quadric_phi(q, phit);
    const bool clipped = sph ? ((zmin > -radius && phit.z < zmin) || (zmax < radius && phit.z > zmax) || phi > phiMax)
                             : (phit.z < zmin || phit.z > zmax || phi > phiMax);
    if (clipped) {
        if (thit == t1) return false;
        if (t1 > maxt) return false;                  // (the others assign thit = t1 before this test: same outcome)
        thit = t1;
        phit = o + d * thit;
        phi = quadric_phi(q, phit);
        if (phit.z < zmin || phit.z > zmax || phi > phiMax) return false;
    }
    t_out = thit;
    return true;
}
// one primitive of a leaf / voxel list: GeometricPrimitive::Intersect(P) (primitive.cpp:103-134) over a triangle or,
// in the EXT kernels, a quadric slot
template <bool EXT>
RT_DEV bool prim_test(const DevScene &sc, unsigned prim, V3 o, V3 d, float mint, float maxt, float &t, float &b1, float &b2) {
    const DevTri RT_G *gt = RT_GPTR(const DevTri, sc.tris) + prim;
    const float4 q0 = gt->q0, q1 = gt->q1, q2 = gt->q2;
    if (EXT && (__float_as_uint(q2.y) & RT_PRIM_QUADRIC)) { b1 = 0.f; b2 = 0.f; return quadric_test(sc, __float_as_uint(q0.x), o, d, mint, maxt, t); }
    return tri_test(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), mk3(q1.z, q1.w, q2.x), o, d, mint, maxt, t, b1, b2);
}

// ---- todo stack --------------------------------------------------------------------------------------------------
// The LDS ring holds the TOP RT_STACK_LDS entries (slot = index mod RT_STACK_LDS); when it is full the OLDEST entry moves to
// HBM, and a pop below the ring's base reads that entry back.  Pushes and pops cluster at the top of the stack, so a deep
// tree (depth 34 at 1M triangles) costs a handful of HBM round trips per ray instead of one per push/pop beyond entry 12.
template <bool COUNT, int NS = RT_STACK_LDS>
RT_DEV void stack_push(Trav &tv, uint2 e, uint2 RT_L *lds_stack, uint2 RT_G *spill, unsigned n_threads, unsigned gtid, TravCounters &cnt) {
    if (tv.sp - tv.sbase == NS) {
        const volatile uint2 RT_L *slot = (const volatile uint2 RT_L *)lds_stack + (unsigned(tv.sbase) % NS) * RT_BLOCK + threadIdx.x;
        uint2 old; old.x = slot->x; old.y = slot->y;
        spill[size_t(tv.sbase) * n_threads + gtid] = old;
        ++tv.sbase;
        if (COUNT) ++cnt.spills;
    }
    lds_stack[(unsigned(tv.sp) % NS) * RT_BLOCK + threadIdx.x] = e;
    ++tv.sp;
}
template <int NS = RT_STACK_LDS>
RT_DEV uint2 stack_pop(Trav &tv, const uint2 RT_L *lds_stack, const uint2 RT_G *spill, unsigned n_threads, unsigned gtid) {
    --tv.sp;
    // always a ds_read_b64; the (rare) spilled entry overrides it -- written this way so that the two loads are
    // not merged into one FLAT load through a selected generic pointer
    const volatile uint2 RT_L *slot = (const volatile uint2 RT_L *)lds_stack + (unsigned(tv.sp) % NS) * RT_BLOCK + threadIdx.x;
    uint2 e; e.x = slot->x; e.y = slot->y;             // volatile: keeps the LDS read a ds_read
    if (tv.sp < tv.sbase) { e = spill[size_t(tv.sp) * n_threads + gtid]; tv.sbase = tv.sp; }
    return e;
}

// start a traversal: slab-clip against the tree bounds (geometry.cpp:51-68, NaN-preserving ternaries)
RT_DEV void trav_begin(Trav &tv, const DevScene &sc, const Ray &r, bool any) {
    tv.o = r.o; tv.d = r.d; tv.mint = r.mint; tv.maxt = r.maxt; tv.any = any;
    tv.hit_prim = -1; tv.b1 = 0.f; tv.b2 = 0.f; tv.sp = 0; tv.sbase = 0; tv.node = 0; tv.at_leaf = false; tv.li = tv.ln_ = tv.ly = 0;
    float t0 = r.mint, t1 = r.maxt;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float invRayDir = 1.f / comp(r.d, i);
        float tNear = (sc.bounds[i] - comp(r.o, i)) * invRayDir;
        float tFar = (sc.bounds[3 + i] - comp(r.o, i)) * invRayDir;
        if (tNear > tFar) { float tmp = tNear; tNear = tFar; tFar = tmp; }
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) ok = false;     // reference returns at the first failing slab; later slabs cannot un-fail it
    }
    tv.tmin = t0; tv.tmax = t1;
    tv.cx = sc.root_x; tv.cy = sc.root_y;
    tv.inv = mk3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
    tv.active = ok && sc.n_tris > 0;
}


// ---- uniform grid: GridAccel::Intersect / IntersectP (reference accelerators/grid.cpp:224-286, :331-390) -------
RT_DEV float arr3(const float *a, int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }
RT_DEV int arr3i(const int *a, int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }
RT_DEV void grid_begin(Trav &tv, const DevScene &sc, const Ray &r, bool any) {
    tv.o = r.o; tv.d = r.d; tv.mint = r.mint; tv.maxt = r.maxt; tv.any = any;
    tv.hit_prim = -1; tv.b1 = 0.f; tv.b2 = 0.f; tv.sp = 0; tv.sbase = 0; tv.node = 0; tv.at_leaf = false; tv.li = tv.ln_ = tv.ly = 0;
    tv.inv = mk3(0.f); tv.tmin = tv.tmax = 0.f;
    float rayT;
    const V3 pm = r.o + r.d * r.mint;                                        // bounds.Inside(ray(ray.mint))
    const bool inside = pm.x >= sc.bounds[0] && pm.x <= sc.bounds[3] && pm.y >= sc.bounds[1] && pm.y <= sc.bounds[4] &&
                        pm.z >= sc.bounds[2] && pm.z <= sc.bounds[5];
    bool ok = true;
    if (inside) rayT = r.mint;
    else {                                                                    // bounds.IntersectP(ray, &rayT)
        float t0 = r.mint, t1 = r.maxt;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float invRayDir = 1.f / comp(r.d, i);
            float tNear = (sc.bounds[i] - comp(r.o, i)) * invRayDir;
            float tFar = (sc.bounds[3 + i] - comp(r.o, i)) * invRayDir;
            if (tNear > tFar) { float tmp = tNear; tNear = tFar; tFar = tmp; }
            t0 = tNear > t0 ? tNear : t0;
            t1 = tFar < t1 ? tFar : t1;
            if (t0 > t1) ok = false;
        }
        rayT = t0;
    }
    const V3 gi = r.o + r.d * rayT;
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
        const float lo = sc.bounds[axis], w = sc.gwidth[axis], da = comp(r.d, axis), ga = comp(gi, axis);
        int v = int((ga - lo) * sc.ginv_width[axis]);                         // PosToVoxel: Float2Int then Clamp
        v = v < 0 ? 0 : (v > sc.nvox[axis] - 1 ? sc.nvox[axis] - 1 : v);
        tv.gpos[axis] = v;
        if (da >= 0) tv.gnext[axis] = rayT + ((lo + (v + 1) * w) - ga) / da;  // VoxelToPos(Pos+1)
        else tv.gnext[axis] = rayT + ((lo + v * w) - ga) / da;
    }
    tv.active = ok && sc.n_tris > 0;
}

// ---- lock-step ("while-while") form of the same traversals ---------------------------------------------------
// The per-lane order of node visits and triangle tests is exactly that of trav_step / grid_step (so hits, ties and
// counters are unchanged); what changes is how the 64 lanes are interleaved: all lanes first walk interior nodes
// (cheap) until each sits at a leaf, then the expensive ray-triangle tests run with every lane that has a primitive
// left -- instead of one lane's 8-triangle leaf serialising against 63 lanes doing 20-instruction plane tests.
// one primitive of the current leaf / voxel list
template <bool COUNT, bool GRID, bool EXT>
RT_DEV void leaf_test_one(Trav &tv, const DevScene &sc, TravCounters &cnt) {
    const bool single = !GRID && tv.ln_ == 1;
    const unsigned prim = single ? tv.ly : RT_GPTR(const unsigned, sc.leaf_refs)[tv.ly + tv.li];
    ++tv.li;
    if (COUNT && !single) ++cnt.leaf_refs;
    if (COUNT) ++cnt.tris;
    float t, b1, b2;
    if (prim_test<EXT>(sc, prim, tv.o, tv.d, tv.mint, tv.maxt, t, b1, b2)) {
        if (tv.any) { tv.hit_prim = 0; tv.active = false; return; }
        tv.maxt = t; tv.hit_prim = int(prim); tv.b1 = b1; tv.b2 = b2;
    }
}
template <int NS = RT_STACK_LDS>
RT_DEV void kd_leaf_done(Trav &tv, const uint2 RT_L *lds_stack, const uint2 RT_G *spill, unsigned n_threads, unsigned gtid) {
    tv.at_leaf = false;
    if (tv.sp > 0) {
        const uint2 e = stack_pop<NS>(tv, lds_stack, spill, n_threads, gtid);
        tv.node = e.x; tv.tmin = tv.tmax; tv.tmax = __uint_as_float(e.y);
    } else tv.active = false;
}
template <bool COUNT>
RT_DEV void grid_enter_voxel(Trav &tv, const DevScene &sc, TravCounters &cnt) {
    const uint2 vx = RT_GPTR(const uint2, sc.nodes)[(size_t(tv.gpos[2]) * sc.nvox[1] + tv.gpos[1]) * sc.nvox[0] + tv.gpos[0]];
    if (COUNT) ++cnt.nodes;
    tv.at_leaf = true; tv.li = 0; tv.ln_ = vx.y; tv.ly = vx.x;
}
RT_DEV void grid_voxel_done(Trav &tv, const DevScene &sc) {               // grid.cpp:273-283
    tv.at_leaf = false;
    const int bits = ((tv.gnext[0] < tv.gnext[1]) << 2) + ((tv.gnext[0] < tv.gnext[2]) << 1) + ((tv.gnext[1] < tv.gnext[2]));
    const int stepAxis = (0x00221212 >> (4 * bits)) & 3;
    const float nx = arr3(tv.gnext, stepAxis);
    if (tv.maxt < nx) { tv.active = false; return; }
    const float da = comp(tv.d, stepAxis);
    const int step = da >= 0 ? 1 : -1, out = da >= 0 ? arr3i(sc.nvox, stepAxis) : -1;
    const int np = arr3i(tv.gpos, stepAxis) + step;
    if (np == out) { tv.active = false; return; }
    const float delta = (da >= 0 ? arr3(sc.gwidth, stepAxis) : -arr3(sc.gwidth, stepAxis)) / da;
    if (stepAxis == 0) { tv.gpos[0] = np; tv.gnext[0] = nx + delta; }
    else if (stepAxis == 1) { tv.gpos[1] = np; tv.gnext[1] = nx + delta; }
    else { tv.gpos[2] = np; tv.gnext[2] = nx + delta; }
}

// ---- pooled leaf tests ----------------------------------------------------------------------------------------
// In a lock-step round the lanes sit in leaves of different sizes (Cornell: 1..9 triangles) and about half of the
// lanes have no live ray at all, so "one triangle per lane per iteration" keeps the wave busy for max(n) iterations
// at ~25 % lane use.  Here the wave pools its (ray, triangle) pairs instead: T = sum(n) tests are dealt to the 64 lanes
// in chunks of 64, whatever lane owns the ray.  A worker lane pulls the owner's ray with ds_bpermute, runs the
// reference's test against the owner's maxt as it was when the leaf was entered, and posts a hit with one LDS
// atomic-min on the owner's 64-bit key {ordered(t), ~k}: the minimum is exactly what the sequential loop
// (trianglemesh.cpp:213-246 under primitive.cpp:120) ends with -- smallest t, the LATER list position on ties
// (a later hit with t == maxt is accepted).  For IntersectP rays the key is k: the first position that hits, which
// also gives the number of tests the sequential loop would have made.  Per-lane visit order, hits, ties and
// counters are unchanged; only the interleaving across lanes differs.
struct PoolLds {
    unsigned long long RT_L *key;   // [64] per wave: best hit of the ray owned by lane i
    float4 RT_L *res;               // [64] {t, b1, b2, prim} of that hit
    unsigned RT_L *head;            // [64] slot -> owner+1 at the first slot of each owner's run
    float4 RT_L *ray;               // [64][3] {o, mint} {d, maxt} {leaf list, n | any << 31, first slot, -} of the ray owned by lane i
};
RT_DEV unsigned ordered_bits(float f) {
    const unsigned b = __float_as_uint(f);
    return b ^ (unsigned(int(b) >> 31) | 0x80000000u);
}
// inclusive wave64 scans on the DPP network (row_shr within rows of 16, then row_bcast:15 / :31 across rows): 6 VALU ops
#define RT_DPP_STEP(OP, CTRL, ROWMASK) { const unsigned u = unsigned(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); v = OP; }
RT_DEV unsigned wave_scan_add(unsigned v) {
    RT_DPP_STEP(v + u, 0x111, 0xf) RT_DPP_STEP(v + u, 0x112, 0xf) RT_DPP_STEP(v + u, 0x114, 0xf) RT_DPP_STEP(v + u, 0x118, 0xf)
    RT_DPP_STEP(v + u, 0x142, 0xa) RT_DPP_STEP(v + u, 0x143, 0xc)
    return v;
}
RT_DEV unsigned wave_scan_max(unsigned v) {
    RT_DPP_STEP(u > v ? u : v, 0x111, 0xf) RT_DPP_STEP(u > v ? u : v, 0x112, 0xf) RT_DPP_STEP(u > v ? u : v, 0x114, 0xf)
    RT_DPP_STEP(u > v ? u : v, 0x118, 0xf) RT_DPP_STEP(u > v ? u : v, 0x142, 0xa) RT_DPP_STEP(u > v ? u : v, 0x143, 0xc)
    return v;
}
#undef RT_DPP_STEP
template <bool COUNT, bool GRID, bool EXT>
RT_DEV void leaf_phase_pooled(Trav &tv, bool need, const DevScene &sc, PoolLds pl, TravCounters &cnt) {
    const int lane = int(__lane_id());
    const unsigned n = need ? tv.ln_ : 0u;
    const unsigned incl = wave_scan_add(n), start = incl - n;
    const unsigned T = unsigned(__builtin_amdgcn_readlane(int(incl), 63));
    // the owner's ray, leaf list and first slot, where any worker lane can read them (3 x ds_write_b128 per round,
    // 3 x ds_read_b128 per chunk instead of a dozen ds_bpermutes)
    pl.key[lane] = tv.any ? ~0ull : ((unsigned long long)ordered_bits(tv.maxt + 0.f) << 32) | 0xFFFFFFFFull;
    pl.ray[3 * lane] = make_float4(tv.o.x, tv.o.y, tv.o.z, tv.mint);
    pl.ray[3 * lane + 1] = make_float4(tv.d.x, tv.d.y, tv.d.z, tv.maxt);
    pl.ray[3 * lane + 2] = make_float4(__uint_as_float(tv.ly), __uint_as_float(tv.ln_ | (tv.any ? 0x80000000u : 0u)), __uint_as_float(start), 0.f);
    for (unsigned c0 = 0; c0 < T; c0 += 64) {
        pl.head[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (n && start < c0 + 64 && start + n > c0) pl.head[(start > c0 ? start : c0) - c0] = unsigned(lane) + 1u;
        __builtin_amdgcn_wave_barrier();
        const unsigned h = wave_scan_max(pl.head[lane]);
        const unsigned w = c0 + unsigned(lane);
        const bool work = w < T;
        const int owner = work ? int(h) - 1 : lane;
        bool hit = false, oany = false;
        unsigned long long key = 0;
        float t = 0.f, b1 = 0.f, b2 = 0.f;
        unsigned prim = 0;
        if (work) {
            const float4 r0 = pl.ray[3 * owner], r1 = pl.ray[3 * owner + 1], r2 = pl.ray[3 * owner + 2];
            const unsigned oly = __float_as_uint(r2.x), opk = __float_as_uint(r2.y), k = w - __float_as_uint(r2.z);
            oany = (opk & 0x80000000u) != 0;
            const bool osingle = !GRID && (opk & 0x7fffffffu) == 1u;
            prim = osingle ? oly : RT_GPTR(const unsigned, sc.leaf_refs)[oly + k];
            hit = prim_test<EXT>(sc, prim, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), r0.w, r1.w, t, b1, b2);
            if (hit) {
                key = oany ? (unsigned long long)k : ((unsigned long long)ordered_bits(t + 0.f) << 32) | (unsigned long long)(0xFFFFFFFEu - k);
                atomicMin((unsigned long long *)(pl.key + owner), key);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (hit && !oany) {                                   // the one lane holding the owner's current minimum publishes it
            if (pl.key[owner] == key) pl.res[owner] = make_float4(t, b1, b2, __uint_as_float(prim));
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (need) {
        const unsigned long long kf = pl.key[lane];
        unsigned tested = tv.ln_;
        if (tv.any) {
            if (kf != ~0ull) { tv.hit_prim = 0; tv.active = false; tested = unsigned(kf) + 1u; }   // kdtree.cpp:432-434
        } else if (unsigned(kf) != 0xFFFFFFFFu) {
            const float4 r = pl.res[lane];
            tv.maxt = r.x; tv.b1 = r.y; tv.b2 = r.z; tv.hit_prim = int(__float_as_uint(r.w));       // primitive.cpp:120
        }
        if (COUNT) { cnt.tris += tested; if (GRID || tv.ln_ != 1) cnt.leaf_refs += tested; }
        tv.li = tv.ln_;
    }
}
template <bool COUNT, int NS, bool LEAF_ORDER>
RT_DEV void kd_step_flat(Trav &tv, bool desc, const DevScene &sc, uint2 RT_L *lds_stack, uint2 RT_G *spill, unsigned n_threads, unsigned gtid, TravCounters &cnt);
// lock-step round with the leaf phase pooled when that is cheaper than max(n) per-lane iterations
template <bool COUNT, int ACCEL, bool EXT>
RT_DEV void accel_round_pooled(Trav &tv, bool mine, const DevScene &sc, uint2 RT_L *lds_stack, uint2 RT_G *spill, unsigned n_threads,
                               unsigned gtid, TravCounters &cnt, PoolLds pl) {
    RT_PFT(unsigned long long t0 = __builtin_readcyclecounter();)
    if (ACCEL == RT_ACCEL_GRID) {
        if (mine && tv.active && !tv.at_leaf) grid_enter_voxel<COUNT>(tv, sc, cnt);
    } else {
        for (;;) {
            const bool desc = mine && tv.active && !tv.at_leaf;
            if (!__any(desc)) break;
            kd_step_flat<COUNT, RT_STACK_LDS, false>(tv, desc, sc, lds_stack, spill, n_threads, gtid, cnt);
        }
    }
    RT_PFT(unsigned long long t1 = __builtin_readcyclecounter(); cnt.c_desc += t1 - t0;)
    const bool need = mine && tv.active && tv.at_leaf;
    const unsigned n = need ? tv.ln_ : 0u;
    // max(n) and sum(n) from ballots (exact while max(n) <= 8; beyond that pooling wins anyway)
    unsigned m = 0, T = 0;
#pragma unroll
    for (unsigned k = 1; k <= 8; ++k) { const unsigned long long b = __ballot(n >= k); if (b) { m = k; T += unsigned(__popcll(b)); } }
    const unsigned chunks = (T + 63u) >> 6;
    if (m >= 8 || (m >= 2 && chunks * 7u + 3u < m * 5u)) {
        RT_PFT(++cnt.n_pooled; cnt.n_chunks += chunks;)
        leaf_phase_pooled<COUNT, ACCEL == RT_ACCEL_GRID, EXT>(tv, need, sc, pl, cnt);
    } else
        while (__any(mine && tv.active && tv.at_leaf && tv.li < tv.ln_)) {
            RT_PFT(++cnt.n_iter;)
            if (mine && tv.active && tv.at_leaf && tv.li < tv.ln_) leaf_test_one<COUNT, ACCEL == RT_ACCEL_GRID, EXT>(tv, sc, cnt);
        }
    RT_PFT(cnt.c_leaf += __builtin_readcyclecounter() - t1;)
    if (mine && tv.active && tv.at_leaf) {
        if (ACCEL == RT_ACCEL_GRID) grid_voxel_done(tv, sc); else kd_leaf_done(tv, lds_stack, spill, n_threads, gtid);
    }
}

// ---- flat (select-style) traversal steps and the round built from them: used by the trace kernel of the queue pipeline
// (rt_pipeline.h) and by the megakernel's trav_mode 2 ----
#ifndef RT_TRACE_DSTEPS
#define RT_TRACE_DSTEPS 2         // steps (of up to two levels each) a descending lane may take per round before the leaf phase gets its turn
                                  // (round 3, two-level steps: 2 beats 4 by 7 % on the 1 M-triangle path frame, 5 % on C3)
#endif
#ifndef RT_TRACE_LEAF_MIN
#define RT_TRACE_LEAF_MIN 24      // keep testing primitives while at least this many lanes have one left (the megakernel takes DevFrame::leaf_min: 8 on tiny trees)
#endif

// entering a leaf of the flat traversal: its first entry and the cursor from the leaf node's two words (encoding: RT_LE_* above)
RT_DEV void leaf_cursor_enter(Trav &tv, bool enter, unsigned wx, unsigned wy) {
    tv.ly = enter ? ((wx >> 2) | (wy & ~RT_LE_POS)) : tv.ly;
    tv.ln_ = enter ? ((wy & RT_LE_POS) << ((wy >> 30) & 1u)) : tv.ln_;     // (a list's cursor is stored halved)
    tv.li = enter ? (wy & RT_LE_MORE) : tv.li;              // read by the counting twins only (the reference walks a list there: leaf_refs); dead elsewhere
}

// ---- the trace kernel's own traversal steps ------------------------------------------------------------------------------
// Same semantics as kd_descend / leaf_test_one / kd_leaf_done of rt_traverse.h (KdTreeAccel::Intersect / IntersectP,
// kdtree.cpp:313-488; Triangle::Intersect, trianglemesh.cpp:213-246), written as straight-line code with selects: every
// state variable is updated by ONE predicated assignment at the end of a step instead of inside nested divergent branches
// with early returns.  The branchy form costs ~30 register copies per step (the structurizer's phi moves); at 135 node
// visits per ray the trace kernel was bound by VALU issue, not by memory.
template <bool COUNT, int NS, bool LEAF_ORDER>
RT_DEV void kd_step_flat(Trav &tv, bool desc, const DevScene &sc, uint2 RT_L *lds_stack, uint2 RT_G *spill, unsigned n_threads, unsigned gtid, TravCounters &cnt) {
    const bool dead = desc && !tv.any && tv.maxt < tv.tmin;                    // kdtree.cpp:330
    const bool go = desc && !dead;
    uint2 nd = make_uint2(3u, 0u);
    if (go) nd = RT_GPTR(const uint2, LEAF_ORDER ? sc.tnodes : sc.nodes)[tv.node];
    if (COUNT) cnt.nodes += go ? 1u : 0u;
    const unsigned axis = nd.x & 3u;
    const bool leaf = axis == 3u;
    const float split = __uint_as_float(nd.x);                                 // perturbed split, B10
    const float oa = comp(tv.o, int(axis)), da = comp(tv.d, int(axis)), ia = comp(tv.inv, int(axis));   // by value: stays in registers
    const float tplane = (split - oa) * ia;
    const bool belowFirst = (oa < split) | ((oa == split) & (da >= 0.f));
    const unsigned below = tv.node + 1u, above = nd.y;
    const unsigned first = belowFirst ? below : above, second = belowFirst ? above : below;
    const bool only_first = tplane > tv.tmax || tplane <= 0.f;
    const bool only_second = !only_first && tplane < tv.tmin;
    const bool interior = go && !leaf;
    const bool both = interior && !only_first && !only_second;
    if (both) stack_push<COUNT, NS>(tv, make_uint2(second, __float_as_uint(tv.tmax)), lds_stack, spill, n_threads, gtid, cnt);
    tv.node = interior ? (only_second ? second : first) : tv.node;
    tv.tmax = both ? tplane : tv.tmax;
    const bool enter = go && leaf;
    tv.at_leaf = enter ? true : tv.at_leaf;
    if (LEAF_ORDER) leaf_cursor_enter(tv, enter, nd.x, nd.y);
    else {
        tv.li = enter ? 0u : tv.li;
        tv.ln_ = enter ? (nd.x >> 2) : tv.ln_;
        tv.ly = enter ? nd.y : tv.ly;
    }
    tv.active = dead ? false : tv.active;
}
template <bool COUNT>
RT_DEV void leaf_test_flat(Trav &tv, bool leafw, const DevScene &sc, TravCounters &cnt) {
    // one record per primitive (DevScene::ltris) at the position the lane's current entry names
    const unsigned cur = tv.ly;
    float4 q0, q1, q2; undef_f4(q0); undef_f4(q1); undef_f4(q2);
    if (leafw) {
        const float4 RT_G *gt = RT_GPTR(const float4, sc.ltris) + (cur & RT_LE_POS);
        q0 = gt[0]; q1 = gt[1]; q2 = gt[2];
    }
    if (sc.leaf_runs) {
        // tiny scenes (scene-uniform, a scalar branch): every leaf owns a run of consecutive records, the cursor counts what is left of it
        const unsigned nxt = tv.ln_ > 1u ? cur + RT_TRI_STRIDE : RT_LE_NONE;
        tv.ly = leafw ? nxt : tv.ly;
        tv.ln_ -= leafw ? 1u : 0u;
    } else {
        // a leaf of three or more asks for the entry of its NEXT primitive in the same batch of loads
        const bool list = leafw && cur >= (RT_LE_MORE | RT_LE_LIST);                 // (leafw: cur is not RT_LE_NONE)
        unsigned nx; asm("" : "=v"(nx));
        if (list) nx = RT_GPTR(const unsigned, sc.lrefs)[tv.ln_];
        const unsigned nxt = list ? nx : (int(cur) < 0 ? tv.ln_ : RT_LE_NONE);   // (a leaf of two: the cursor is the second entry)
        tv.ly = leafw ? nxt : tv.ly;
        tv.ln_ += list ? 1u : 0u;
    }
    const unsigned prim = __float_as_uint(q2.w);
    if (COUNT) { cnt.tris += leafw ? 1u : 0u; cnt.leaf_refs += (leafw && tv.li != 0u) ? 1u : 0u; }      // tv.li (counting twins only): the leaf has more than one primitive
    const V3 p1 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
    const V3 s1 = cross3(tv.d, e2);
    const float divisor = dot3(s1, e1);
    const float invDivisor = 1.f / divisor;
    const V3 dd = tv.o - p1;
    const float b1 = dot3(dd, s1) * invDivisor;
    const V3 s2 = cross3(dd, e1);
    const float b2 = dot3(tv.d, s2) * invDivisor;
    const float t = dot3(e2, s2) * invDivisor;
    const bool miss = (divisor == 0.f) | (b1 < 0.f) | (b1 > 1.f) | (b2 < 0.f) | (b1 + b2 > 1.f) | (t < tv.mint) | (t > tv.maxt);
    const bool hit = leafw && !miss;
    const bool stop = hit && tv.any;                                           // kdtree.cpp:432-434
    const bool keep = hit && !tv.any;                                          // primitive.cpp:120
    tv.hit_prim = stop ? 0 : (keep ? int(prim) : tv.hit_prim);
    tv.maxt = keep ? t : tv.maxt;
    tv.b1 = keep ? b1 : tv.b1;
    tv.b2 = keep ? b2 : tv.b2;
    tv.active = stop ? false : tv.active;
}
template <int NS>
RT_DEV void kd_pop_flat(Trav &tv, bool done, const uint2 RT_L *lds_stack, const uint2 RT_G *spill, unsigned n_threads, unsigned gtid) {
    const bool pop = done && tv.sp > 0;
    uint2 e = make_uint2(0u, 0u);
    if (pop) e = stack_pop<NS>(tv, lds_stack, spill, n_threads, gtid);
    tv.node = pop ? e.x : tv.node;
    tv.tmin = pop ? tv.tmax : tv.tmin;
    tv.tmax = pop ? __uint_as_float(e.y) : tv.tmax;
    tv.at_leaf = done ? false : tv.at_leaf;
    tv.active = (done && !pop) ? false : tv.active;
}

// ---- sibling-pair form of the node step and the pop (DevScene::tpairs) -----------------------------------------------------
// Stack entry = {far child's node words, tmax}: 8 + 4 bytes in two LDS planes [NS][RT_BLOCK]; the oldest entries spill to HBM as
// uint4.  Visit order, tie rules and counters are those of kd_step_flat (kdtree.cpp:313-488); what changes is what is fetched.
struct PairStack { uint2 RT_L *xy; float RT_L *tm; uint4 RT_G *spill; };
// push {far child's words, tmax}; when the ring is full its oldest entry moves to HBM
template <bool COUNT, int NS>
RT_DEV void kdp_push(Trav &tv, PairStack st, unsigned sx, unsigned sy, float tmax, unsigned n_threads, unsigned gtid, TravCounters &cnt) {
    if (tv.sp - tv.sbase == NS) {
        const unsigned o = (unsigned(tv.sbase) % NS) * RT_BLOCK + threadIdx.x;
        const volatile uint2 RT_L *ox = (const volatile uint2 RT_L *)st.xy + o;
        const volatile float RT_L *ot = (const volatile float RT_L *)st.tm + o;
        st.spill[size_t(tv.sbase) * n_threads + gtid] = make_uint4(ox->x, ox->y, __float_as_uint(*ot), 0u);
        ++tv.sbase;
        if (COUNT) ++cnt.spills;
    }
    const unsigned w = (unsigned(tv.sp) % NS) * RT_BLOCK + threadIdx.x;
    st.xy[w] = make_uint2(sx, sy); st.tm[w] = tmax;
    ++tv.sp;
}
// One step = one gather round trip = up to TWO levels of the tree.  Everything that decides which child of the current node P the
// traversal continues in -- P's split (in hand), the ray, [tmin, tmax] -- is known BEFORE P's children arrive, and the pair layout
// (pair_blocks_order / pair_blocks_fill, rt_scene.hip) puts the pairs of an "owner" node's interior children right behind the owner's own pair
// ({P, below(P), above(P)}: at most 48 bytes, never across a 64-byte boundary; bits 30 / 31 of the owner's word 1 say which of them
// exist).  So the step asks for pair(P) and pair(chosen child) together, and when they arrive takes the reference's decisions for P
// and for that child back to back (kdtree.cpp:340-365, same comparisons, same push order).  A node that is not an owner (flags 0: a
// member of its parent's block, reached through a pop) takes the one-level form of the same code.
template <bool COUNT, int NS>
RT_DEV void kdp_step(Trav &tv, bool desc, const DevScene &sc, PairStack st, unsigned n_threads, unsigned gtid, TravCounters &cnt) {
    const bool dead = desc && !tv.any && tv.maxt < tv.tmin;                    // kdtree.cpp:330
    const bool go = desc && !dead;
    if (COUNT) cnt.nodes += go ? 1u : 0u;
    tv.active = dead ? false : tv.active;
    const bool leaf = (tv.cx & 3u) == 3u;                                      // only ever a root that is a leaf
    const unsigned axis = tv.cx & 3u;
    const bool interior = go && !leaf;
    // ---- level 1: decided from the words in hand
    const float split = __uint_as_float(tv.cx);                                // perturbed split, B10
    const float oa = comp(tv.o, int(axis)), da = comp(tv.d, int(axis)), ia = comp(tv.inv, int(axis));
    const float tplane = (split - oa) * ia;
    const bool belowFirst = (oa < split) | ((oa == split) & (da >= 0.f));        // bitwise: lane masks on the scalar unit, no divergent branch
    const bool only_first = (tplane > tv.tmax) | (tplane <= 0.f);
    const bool only_second = !only_first & (tplane < tv.tmin);
    const bool both = interior & !only_first & !only_second;
    const bool c_above = !(belowFirst ^ only_second);                          // the child the traversal continues in: belowFirst ? only_second : !only_second
    const unsigned idx = tv.cy & 0x3fffffffu, fb = (tv.cy >> 30) & 1u, fa = tv.cy >> 31;
    const bool two = interior && (c_above ? fa : fb) != 0u;                    // that child's pair sits in this node's block
    uint4 A, B; undef_u4(A); undef_u4(B);
    if (interior) {
        const uint4 RT_G *p = RT_GPTR(const uint4, sc.tpairs) + idx;
        A = p[0];
        if (two) B = p[1u + (c_above ? fb : 0u)];
    }
    const unsigned c_x = c_above ? A.z : A.x, c_y = c_above ? A.w : A.y;
    const unsigned f_x = c_above ? A.x : A.z, f_y = c_above ? A.y : A.w;
    if (both) kdp_push<COUNT, NS>(tv, st, f_x, f_y, tv.tmax, n_threads, gtid, cnt);
    const float tmax1 = both ? tplane : tv.tmax;
    // ---- level 2: the chosen child (interior, its pair is B)
    const unsigned axis2 = c_x & 3u;
    const float split2 = __uint_as_float(c_x);
    const float oa2 = comp(tv.o, int(axis2)), da2 = comp(tv.d, int(axis2)), ia2 = comp(tv.inv, int(axis2));
    const float tplane2 = (split2 - oa2) * ia2;
    const bool belowFirst2 = (oa2 < split2) | ((oa2 == split2) & (da2 >= 0.f));
    const bool only_first2 = (tplane2 > tmax1) | (tplane2 <= 0.f);
    const bool only_second2 = !only_first2 & (tplane2 < tv.tmin);
    const bool both2 = two & !only_first2 & !only_second2;
    const bool g_above = !(belowFirst2 ^ only_second2);
    const unsigned g_x = g_above ? B.z : B.x, g_y = g_above ? B.w : B.y;
    const unsigned h_x = g_above ? B.x : B.z, h_y = g_above ? B.y : B.w;
    if (both2) kdp_push<COUNT, NS>(tv, st, h_x, h_y, tmax1, n_threads, gtid, cnt);
    if (COUNT) cnt.nodes += two ? 1u : 0u;
    tv.cx = interior ? (two ? g_x : c_x) : tv.cx;
    tv.cy = interior ? (two ? g_y : c_y) : tv.cy;
    tv.tmax = interior ? (both2 ? tplane2 : tmax1) : tv.tmax;
    // The chosen child's words are in hand, so a leaf child is entered in this very step (the reference's next iteration re-checks
    // maxt < tmin with unchanged values, kdtree.cpp:330, and then is in the leaf); `leaf` itself is only ever a root that is a leaf.
    const bool enter = go && (tv.cx & 3u) == 3u;
    if (COUNT) cnt.nodes += (interior && enter) ? 1u : 0u;
    tv.at_leaf = enter ? true : tv.at_leaf;
    leaf_cursor_enter(tv, enter, tv.cx, tv.cy);
}
template <bool COUNT, int NS>
RT_DEV void kdp_pop(Trav &tv, bool done, PairStack st, unsigned n_threads, unsigned gtid, TravCounters &cnt) {
    const bool pop = done && tv.sp > 0;
    unsigned ex, ey; float et; asm("" : "=v"(ex), "=v"(ey), "=v"(et));
    if (pop) {
        --tv.sp;
        const unsigned r = (unsigned(tv.sp) % NS) * RT_BLOCK + threadIdx.x;
        const volatile uint2 RT_L *px = (const volatile uint2 RT_L *)st.xy + r;
        const volatile float RT_L *pt = (const volatile float RT_L *)st.tm + r;
        ex = px->x; ey = px->y; et = *pt;
        if (tv.sp < tv.sbase) { const uint4 e = st.spill[size_t(tv.sp) * n_threads + gtid]; ex = e.x; ey = e.y; et = __uint_as_float(e.z); tv.sbase = tv.sp; }
    }
    tv.cx = pop ? ex : tv.cx;
    tv.cy = pop ? ey : tv.cy;
    tv.tmin = pop ? tv.tmax : tv.tmin;
    tv.tmax = pop ? et : tv.tmax;
    // the popped node's words are in hand: the reference's next iteration checks maxt < tmin (kdtree.cpp:330) and, for a leaf, is in it
    const bool dead = pop && !tv.any && tv.maxt < tv.tmin;
    const bool enter = pop && !dead && (ex & 3u) == 3u;
    if (COUNT) cnt.nodes += enter ? 1u : 0u;
    tv.at_leaf = done ? enter : tv.at_leaf;
    leaf_cursor_enter(tv, enter, ex, ey);
    tv.active = (done && (!pop || dead)) ? false : tv.active;
}

template <bool COUNT, int ACCEL, bool EXT, int NS, bool PAIRS_OK = true, int DSTEPS = RT_TRACE_DSTEPS>
RT_DEV void trace_round(Trav &tv, bool busy, const DevScene &sc, uint2 RT_L *lds_stack, float RT_L *lds_tm, uint2 RT_G *spill, unsigned n_threads, unsigned gtid, TravCounters &cnt,
                        int leaf_min = RT_TRACE_LEAF_MIN) {
    constexpr bool PAIRS = PAIRS_OK && ACCEL != RT_ACCEL_GRID && !EXT;
    constexpr bool CURSOR = ACCEL != RT_ACCEL_GRID && !EXT;                    // the leaf cursor walks entries (one record per primitive), not [li, ln)
    const PairStack pst = {lds_stack, lds_tm, (uint4 RT_G *)spill};
    if (ACCEL == RT_ACCEL_GRID) {
        if (busy && tv.active && !tv.at_leaf) grid_enter_voxel<COUNT>(tv, sc, cnt);
    } else if (PAIRS) {
#pragma unroll 1
        for (int k = 0; k < DSTEPS; ++k) {
            const bool desc = busy && tv.active && !tv.at_leaf;
            if (!__any(desc)) break;
            kdp_step<COUNT, NS>(tv, desc, sc, pst, n_threads, gtid, cnt);
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < DSTEPS; ++k) {
            const bool desc = busy && tv.active && !tv.at_leaf;
            if (!__any(desc)) break;
            kd_step_flat<COUNT, NS, !EXT>(tv, desc, sc, lds_stack, spill, n_threads, gtid, cnt);
        }
    }
#pragma unroll 1
    for (;;) {
        const bool leafw = busy && tv.active && tv.at_leaf && (CURSOR ? tv.ly != RT_LE_NONE : tv.li < tv.ln_);
        const int nl = __popcll(__ballot(leafw));
        if (nl == 0) break;
        if (ACCEL == RT_ACCEL_GRID || EXT) { if (leafw) leaf_test_one<COUNT, ACCEL == RT_ACCEL_GRID, EXT>(tv, sc, cnt); }
        else leaf_test_flat<COUNT>(tv, leafw, sc, cnt);
        if (nl < leaf_min) break;
    }
    const bool done = busy && tv.active && tv.at_leaf && (CURSOR ? tv.ly == RT_LE_NONE : tv.li >= tv.ln_);
    if (ACCEL == RT_ACCEL_GRID) { if (done) grid_voxel_done(tv, sc); }
    else if (PAIRS) { if (__any(done)) kdp_pop<COUNT, NS>(tv, done, pst, n_threads, gtid, cnt); }
    else kd_pop_flat<NS>(tv, done, lds_stack, spill, n_threads, gtid);
}

// accelerator dispatch (compile-time)
template <int ACCEL>
RT_DEV void accel_begin(Trav &tv, const DevScene &sc, const Ray &r, bool any) {
    if (ACCEL == RT_ACCEL_GRID) grid_begin(tv, sc, r, any); else trav_begin(tv, sc, r, any);
}
}  // namespace rt
