// rt_internal.h -- declarations shared by the host-side translation units of libpbrt_hip.so
#pragma once
#include "../../../include/pbrt_hip.h"
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

namespace rt {

struct Node { uint32_t x, y; };   // 8-byte flattened kd node (see kd_build.cpp header)

// A vector whose resize() leaves the new elements uninitialised: multi-gigabyte arrays (2 GB of nodes and 4 GB of leaf references at 10 M triangles,
// 12 GB of leaf-ordered records) are first touched by the threads that fill them, not zero-filled page by page by one thread beforehand.
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U> void construct(U *) noexcept {}
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
typedef std::vector<Node, NoInitAlloc<Node>> NodeVec;
typedef std::vector<uint32_t, NoInitAlloc<uint32_t>> RefVec;

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

}  // namespace rt
