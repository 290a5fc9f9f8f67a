// rt_internal.h -- declarations shared by the host-side translation units of libpbrt_hip.so
#pragma once
#include "../../../include/pbrt_hip.h"
#include <cstdint>
#include <vector>

namespace rt {

struct Node { uint32_t x, y; };   // 8-byte flattened kd node (see kd_build.cpp header)

struct KdTree {
    std::vector<Node> nodes;
    std::vector<uint32_t> leaf_refs;
    float bounds[6];              // lo xyz, hi xyz
    int max_depth = 0;
    double build_seconds = 0;
};

struct GridAccelData {
    std::vector<Node> voxels;         // {offset, count} per voxel, x fastest
    std::vector<uint32_t> refs;
    int nvox[3];
    float width[3], inv_width[3], bounds[6];
    double build_seconds = 0;
};
void build_grid(const float *tri_verts, uint32_t n_tris, GridAccelData &out);

void build_kdtree(const float *tri_verts, uint32_t n_tris, const RtAccelParams &params, KdTree &out);

}  // namespace rt
