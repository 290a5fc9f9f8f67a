// rt_internal.h -- declarations shared by the host-side translation units of libpbrt_hip.so
#pragma once
#include "../../../include/pbrt_hip.h"
#include <cstdint>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

namespace rt {

struct Node { uint32_t x, y; };   // 8-byte flattened kd node (see kd_build.cpp header)

// A vector whose resize() leaves the new elements uninitialised: multi-gigabyte arrays (2 GB of nodes and 4 GB of leaf references at 10 M triangles,
// 3.6 GB of leaf entries) are first touched by the threads that fill them, not zero-filled page by page by one thread beforehand.
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U> void construct(U *) noexcept {}
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
typedef std::vector<Node, NoInitAlloc<Node>> NodeVec;
typedef std::vector<uint32_t, NoInitAlloc<uint32_t>> RefVec;

// Worker threads that are joined on EVERY way out of the scope that started them.  A std::thread that is still joinable when it is destroyed calls
// std::terminate: if starting the k-th worker throws (EAGAIN under thread / ulimit pressure) or an allocation beside the workers throws bad_alloc, the
// workers already running are joined during unwinding and the exception reaches the C boundary (rt_scene_create / rt_kdtree_build return RT_ENOMEM /
// RT_ESTATE) instead of ending the process.
struct ThreadGroup {
    std::vector<std::thread> th;
    template <class F, class... A> void spawn(F &&f, A &&...a) { th.emplace_back(std::forward<F>(f), std::forward<A>(a)...); }
    void join() { for (auto &t : th) if (t.joinable()) t.join(); th.clear(); }
    ~ThreadGroup() { for (auto &t : th) if (t.joinable()) t.join(); }
};

struct KdTree {
    NodeVec nodes;
    RefVec leaf_refs;
    float bounds[6];              // lo xyz, hi xyz
    int max_depth = 0;
    double build_seconds = 0;
};

struct GridAccelData {
    NodeVec voxels;                   // {offset, count} per voxel, x fastest
    RefVec refs;
    int nvox[3];
    float width[3], inv_width[3], bounds[6];
    double build_seconds = 0;
};
void build_grid(const float *tri_verts, uint32_t n_tris, GridAccelData &out);

void build_kdtree(const float *tri_verts, uint32_t n_tris, const RtAccelParams &params, KdTree &out);

// The leaves of a flattened kd-tree as the flat traversal reads them (leaf_layout.cpp; encoding: rt_leaf_entries.h)
struct LeafLayout {
    NodeVec tnodes;                       // the nodes with leaves in entry form
    RefVec lrefs;                         // entries of the third and later primitives of the leaves
    RefVec slot_prim;                     // record slot -> primitive
    size_t n_slots = 0;
};
bool leaf_cursor_layout(const NodeVec &nodes, const RefVec &leaf_refs, uint32_t n_tris, bool copies, bool runs, LeafLayout &out);

}  // namespace rt
