// kd_build.cpp -- host-side SAH kd-tree construction producing the flattened 8-byte
// node array the HIP traversal kernels consume.
//
// Behavioural contract = KdTreeAccel's constructor and buildTree
// (reference accelerators/kdtree.cpp:141-312), restated -- not copied -- so that for the
// same primitive list and parameters the tree has the same shape: same depth limit
// (kdtree.cpp:159-161 with Log2Int rounding to nearest, core/pbrt.h:575-586,604-613), same
// first-axis rule, same (t, START<END) edge order fed to the same std::sort, same
// "strictly inside" candidate test, same cost formula and empty bonus, same retry /
// bad-refine / 4x-cost leaf rules, same straddler duplication (kdtree.cpp:225-311).
// Node encoding (kdtree.cpp:36-88): word0 = split float whose 2 low mantissa bits are
// overwritten with the axis (interior) or (nPrims<<2)|3 (leaf); word1 = index of the
// above child (the below child is always node+1), or for leaves the single primitive /
// the offset of the leaf's reference list.  A leaf with one primitive stores it inline.
#include "rt_internal.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <atomic>
#include <thread>

namespace rt {
namespace {

struct Edge {
    float t;
    int prim;
    int kind;  // 0 = lower bound ("start"), 1 = upper bound ("end")
    bool operator<(const Edge &o) const { return t == o.t ? kind < o.kind : t < o.t; }
};

struct Box { float lo[3], hi[3]; };

inline int round_to_int(double v) { return int(v + (.5 - 1.4e-11)); }          // pbrt.h:604-613
inline int log2_int(float v) {                                                    // pbrt.h:575-586
    static const float inv_log2 = 1.f / logf(2.f);
    return int(double(logf(v) * inv_log2) + (.5 - 1.4e-11));
}

class Builder {
  public:
    Builder(const float *verts, uint32_t n, const RtAccelParams &p, KdTree &out)
        : nTris(n), prm(p), tree(out) {
        primBoxOwn.resize(n); primBoxPtr = &primBoxOwn; std::vector<Box> &primBox = primBoxOwn;
        for (int a = 0; a < 3; ++a) { tree.bounds[a] = INFINITY; tree.bounds[3 + a] = -INFINITY; }
        for (uint32_t i = 0; i < n; ++i) {
            const float *v = verts + size_t(9) * i;
            Box &b = primBox[i];
            for (int a = 0; a < 3; ++a) {  // Union(BBox(p1,p2), p3), trianglemesh.cpp:205-211
                b.lo[a] = std::min(std::min(v[a], v[3 + a]), v[6 + a]);
                b.hi[a] = std::max(std::max(v[a], v[3 + a]), v[6 + a]);
                tree.bounds[a] = std::min(tree.bounds[a], b.lo[a]);
                tree.bounds[3 + a] = std::max(tree.bounds[3 + a], b.hi[a]);
            }
        }
    }

    // ---- parallel build: the top of the tree is built serially; every subtree with <= cutoff primitives becomes a task
    // that an independent Builder turns into its own node / leaf-reference arrays; stitching them back in depth-first
    // order reproduces the serial build's arrays exactly (each subtree's construction depends only on its arguments).
    struct Task { Box nb; std::vector<int> prims; int depth, bad; uint32_t placeholder; KdTree sub; };
    std::vector<Task> *tasks = nullptr;
    int cutoff = 0;
    // fork-join over the top of the tree: while fork_levels > 0 a node's ABOVE child is built by a forked thread (its own Builder,
    // node array and task list) while this thread builds the below child in place; the forked arrays are then appended with their
    // indices rebased.  Every node's construction still depends only on its own arguments and runs the same std::sort, so the
    // arrays are those of the serial build; the critical path becomes one node per level (10 M primitives: ~30 s -> single digits).
    int fork_levels = 0;
    void build_subtree(const Box &nb, const std::vector<int> &prims, int depth, int bad) {
        const int n = int(prims.size());
        for (int a = 0; a < 3; ++a) edges[a].resize(size_t(2) * std::max(n, 1));
        std::vector<int> below(std::max(n, 1)), above(size_t(depth + 1) * std::max(n, 1));
        split(nb, prims.data(), n, depth, below.data(), above.data(), bad);
    }

    void run() {
        int depth = prm.max_depth;
        if (nTris == 0) depth = 1;                       // empty world: one empty leaf (the reference evaluates log(0) here)
        else if (depth <= 0) depth = round_to_int(8 + 1.3f * log2_int(float(nTris)));
        tree.max_depth = depth;
        for (int a = 0; a < 3; ++a) edges[a].resize(size_t(2) * nTris);
        std::vector<int> below(nTris), above(size_t(depth + 1) * nTris), all(nTris);
        for (uint32_t i = 0; i < nTris; ++i) all[i] = int(i);
        Box root; for (int a = 0; a < 3; ++a) { root.lo[a] = tree.bounds[a]; root.hi[a] = tree.bounds[3 + a]; }
        int threads = prm.build_threads > 0 ? prm.build_threads : int(std::thread::hardware_concurrency());
        if (threads > 64) threads = 64;
        if (threads < 2 || nTris < 20000) {             // serial
            tree.nodes.reserve(size_t(nTris) * 4 + 64);
            split(root, all.data(), int(nTris), depth, below.data(), above.data(), 0);
            return;
        }
        std::vector<Task> tk; tasks = &tk; cutoff = std::max(2048, int(nTris / (8 * threads)));
        fork_levels = 0; for (int t = 1; t < threads && fork_levels < 6; t <<= 1) ++fork_levels;
        split(root, all.data(), int(nTris), depth, below.data(), above.data(), 0);
        fork_levels = 0;
        tasks = nullptr;
        std::vector<int>().swap(above); std::vector<int>().swap(below);
        std::atomic<size_t> next(0);
        auto worker = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= tk.size()) return;
                Builder sub(*this, tk[i].sub);
                sub.build_subtree(tk[i].nb, tk[i].prims, tk[i].depth, tk[i].bad);
                std::vector<int>().swap(tk[i].prims);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
        for (auto &th : pool) th.join();
        // stitch: depth-first walk of the serial top; placeholders expand to their subtree with rebased indices
        std::vector<Node> top; top.swap(tree.nodes);
        std::vector<uint32_t> topRefs; topRefs.swap(tree.leaf_refs);
        size_t total = top.size(), totalRefs = topRefs.size();
        for (const Task &t : tk) { total += t.sub.nodes.size() - 1; totalRefs += t.sub.leaf_refs.size(); }
        tree.nodes.reserve(total); tree.leaf_refs.reserve(totalRefs);
        struct Emit {
            const std::vector<Node> &top; const std::vector<uint32_t> &topRefs; std::vector<Task> &tk; KdTree &out;
            void go(uint32_t i) {
                const Node nd = top[i];
                if (nd.x == 0xFFFFFFFFu) {                                  // placeholder -> whole subtree
                    const Task &t = tk[nd.y];
                    const uint32_t nodeBase = uint32_t(out.nodes.size()), refBase = uint32_t(out.leaf_refs.size());
                    for (const Node &sn : t.sub.nodes) {
                        Node o = sn;
                        if ((sn.x & 3u) != 3u) o.y = sn.y + nodeBase;       // interior: above-child index
                        else if ((sn.x >> 2) > 1) o.y = sn.y + refBase;     // leaf with a reference list
                        out.nodes.push_back(o);
                    }
                    out.leaf_refs.insert(out.leaf_refs.end(), t.sub.leaf_refs.begin(), t.sub.leaf_refs.end());
                    return;
                }
                const uint32_t me = uint32_t(out.nodes.size());
                out.nodes.push_back(nd);
                if ((nd.x & 3u) == 3u) {                                    // leaf built in the serial top
                    const uint32_t np = nd.x >> 2;
                    if (np > 1) { out.nodes[me].y = uint32_t(out.leaf_refs.size()); for (uint32_t k = 0; k < np; ++k) out.leaf_refs.push_back(topRefs[nd.y + k]); }
                    return;
                }
                go(i + 1);                                                  // below child
                out.nodes[me].y = uint32_t(out.nodes.size());
                go(nd.y);                                                   // above child (top-array index)
            }
        } emit{top, topRefs, tk, tree};
        emit.go(0);
    }

    // subtree builder borrowing the parent's primitive boxes
    Builder(const Builder &parent, KdTree &out) : nTris(parent.nTris), prm(parent.prm), tree(out), primBoxPtr(parent.primBoxPtr) {}

  private:
    uint32_t nTris;
    RtAccelParams prm;
    KdTree &tree;
    std::vector<Box> primBoxOwn;
    const std::vector<Box> *primBoxPtr = nullptr;
    std::vector<Edge> edges[3];

    void leaf(uint32_t at, const int *prims, int n) {
        tree.nodes[at].x = (uint32_t(n) << 2) | 3u;
        if (n == 0) tree.nodes[at].y = 0;
        else if (n == 1) tree.nodes[at].y = uint32_t(prims[0]);
        else {
            tree.nodes[at].y = uint32_t(tree.leaf_refs.size());
            for (int i = 0; i < n; ++i) tree.leaf_refs.push_back(uint32_t(prims[i]));
        }
    }

    void split(const Box &nb, const int *prims, int n, int depth, int *below, int *above, int bad) {
        const uint32_t me = uint32_t(tree.nodes.size());
        tree.nodes.push_back(Node{0, 0});
        if (n <= prm.max_prims || depth == 0) { leaf(me, prims, n); return; }
        if (tasks && n <= cutoff) {                       // hand this subtree to the task pool
            Task t; t.nb = nb; t.prims.assign(prims, prims + n); t.depth = depth; t.bad = bad; t.placeholder = me;
            tree.nodes[me].x = 0xFFFFFFFFu; tree.nodes[me].y = uint32_t(tasks->size());
            tasks->push_back(std::move(t));
            return;
        }

        int bestAxis = -1, bestEdge = -1;
        float bestCost = INFINITY;
        const float leafCost = prm.isect_cost * float(n);
        const float d[3] = {nb.hi[0] - nb.lo[0], nb.hi[1] - nb.lo[1], nb.hi[2] - nb.lo[2]};
        const float totalSA = (2.f * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]));
        const float invTotalSA = 1.f / totalSA;
        int axis = (d[0] > d[1] && d[0] > d[2]) ? 0 : ((d[1] > d[2]) ? 1 : 2);

        for (int attempt = 0;; ++attempt) {
            Edge *e = edges[axis].data();
            for (int i = 0; i < n; ++i) {
                const int p = prims[i];
                e[2 * i] = Edge{(*primBoxPtr)[p].lo[axis], p, 0};
                e[2 * i + 1] = Edge{(*primBoxPtr)[p].hi[axis], p, 1};
            }
            std::sort(e, e + 2 * n);
            int nBelow = 0, nAbove = n;
            const int o0 = axis == 0 ? 1 : 0;                              // {1,2},{0,2},{0,1}
            const int o1 = axis == 2 ? 1 : 2;
            for (int i = 0; i < 2 * n; ++i) {
                if (e[i].kind == 1) --nAbove;
                const float t = e[i].t;
                if (t > nb.lo[axis] && t < nb.hi[axis]) {
                    const float belowSA = 2 * (d[o0] * d[o1] + (t - nb.lo[axis]) * (d[o0] + d[o1]));
                    const float aboveSA = 2 * (d[o0] * d[o1] + (nb.hi[axis] - t) * (d[o0] + d[o1]));
                    const float pB = belowSA * invTotalSA, pA = aboveSA * invTotalSA;
                    const float bonus = (nAbove == 0 || nBelow == 0) ? prm.empty_bonus : 0.f;
                    const float cost = prm.trav_cost + prm.isect_cost * (1.f - bonus) * (pB * nBelow + pA * nAbove);
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestEdge = i; }
                }
                if (e[i].kind == 0) ++nBelow;
            }
            if (bestAxis == -1 && attempt < 2) { axis = (axis + 1) % 3; continue; }
            break;
        }
        if (bestCost > leafCost) ++bad;
        if ((bestCost > 4.f * leafCost && n < 16) || bestAxis == -1 || bad == 3) { leaf(me, prims, n); return; }

        const Edge *e = edges[bestAxis].data();
        int n0 = 0, n1 = 0;
        for (int i = 0; i < bestEdge; ++i) if (e[i].kind == 0) below[n0++] = e[i].prim;
        for (int i = bestEdge + 1; i < 2 * n; ++i) if (e[i].kind == 1) above[n1++] = e[i].prim;
        const float ts = e[bestEdge].t;
        uint32_t bits; std::memcpy(&bits, &ts, 4);
        tree.nodes[me].x = (bits & ~3u) | uint32_t(bestAxis);
        Box b0 = nb, b1 = nb;
        b0.hi[bestAxis] = b1.lo[bestAxis] = ts;
        if (tasks && fork_levels > 0 && n1 > cutoff && n0 > cutoff) {
            // above child in a forked thread (own buffers), below child here; then append the forked arrays behind ours
            KdTree sub; std::vector<Task> subTasks;
            std::vector<int> abovePrims(above, above + n1);
            const int levels = fork_levels - 1, cut = cutoff;
            std::thread th([&, levels, cut]() {
                Builder fb(*this, sub);
                fb.tasks = &subTasks; fb.cutoff = cut; fb.fork_levels = levels;
                for (int a = 0; a < 3; ++a) fb.edges[a].resize(size_t(2) * n1);
                std::vector<int> fbelow(n1), fabove(size_t(depth) * n1);
                fb.split(b1, abovePrims.data(), n1, depth - 1, fbelow.data(), fabove.data(), bad);
            });
            const int saved = fork_levels; fork_levels = levels;
            split(b0, below, n0, depth - 1, below, above + n, bad);
            fork_levels = saved;
            th.join();
            const uint32_t nodeBase = uint32_t(tree.nodes.size()), refBase = uint32_t(tree.leaf_refs.size()), taskBase = uint32_t(tasks->size());
            tree.nodes[me].y = nodeBase;
            for (const Node &sn : sub.nodes) {
                Node o = sn;
                if (sn.x == 0xFFFFFFFFu) o.y = sn.y + taskBase;              // placeholder of a pooled subtree
                else if ((sn.x & 3u) != 3u) o.y = sn.y + nodeBase;           // interior: above-child index
                else if ((sn.x >> 2) > 1) o.y = sn.y + refBase;              // leaf of the top with a reference list
                tree.nodes.push_back(o);
            }
            tree.leaf_refs.insert(tree.leaf_refs.end(), sub.leaf_refs.begin(), sub.leaf_refs.end());
            for (Task &t : subTasks) { t.placeholder += nodeBase; tasks->push_back(std::move(t)); }
            return;
        }
        split(b0, below, n0, depth - 1, below, above + n, bad);
        tree.nodes[me].y = uint32_t(tree.nodes.size());
        split(b1, above, n1, depth - 1, below, above + n, bad);
    }
};

}  // namespace

void build_kdtree(const float *tri_verts, uint32_t n_tris, const RtAccelParams &params, KdTree &out) {
    auto t0 = std::chrono::steady_clock::now();
    out.nodes.clear(); out.leaf_refs.clear();
    RtAccelParams p = params;
    if (p.isect_cost == 0 && p.trav_cost == 0 && p.max_prims == 0) {  // all-zero struct => reference defaults
        p.isect_cost = 80; p.trav_cost = 1; p.max_prims = 1; p.max_depth = -1; p.empty_bonus = 0.5f;
    }
    Builder b(tri_verts, n_tris, p, out);
    b.run();
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace rt
