// kd_build.cpp -- host-side SAH kd-tree construction producing the flattened 8-byte
// node array the HIP traversal kernels consume.
//
// Behavioural contract = KdTreeAccel's constructor and buildTree
// (reference accelerators/kdtree.cpp:141-312), restated -- not copied -- so that for the
// same primitive list and parameters the tree has the same shape: same depth limit
// (kdtree.cpp:159-161 with Log2Int rounding to nearest, core/pbrt.h:575-586,604-613), same
// first-axis rule, same (t, START<END) edge order, same "strictly inside" candidate test, same cost
// formula and empty bonus, same retry / bad-refine / 4x-cost leaf rules, same straddler
// duplication (kdtree.cpp:225-311).
// Node encoding (kdtree.cpp:36-88): word0 = split float whose 2 low mantissa bits are
// overwritten with the axis (interior) or (nPrims<<2)|3 (leaf); word1 = index of the
// above child (the below child is always node+1), or for leaves the single primitive /
// the offset of the leaf's reference list.  A leaf with one primitive stores it inline.
//
// Round 5: SORT ONCE.  The reference sorts the 2n bound edges of every node on the axis it tries (kdtree.cpp:246, std::sort on
// (t, START < END)); edges that compare equal keep whatever order that sort leaves them in -- unspecified -- and the order is visible: it
// is the order of a leaf's primitives.  This builder defines it: (t, START < END, primitive number).  Under a total order the sorted edge
// list of a node IS its parent's sorted list with the edges of the primitives that went elsewhere removed (a primitive's bounds are never
// clipped to the node, kdtree.cpp:169-173), so the three lists are sorted ONCE at the root (in parallel) and every node below FILTERS its
// parent's lists: O(n) per node and axis instead of O(n log n), and nothing serial at the top of the tree but two linear passes.  The
// sorting form of the same build (build_threads = -1: every node sorts, as the reference does, with the same total order) is kept as
// the check: tests assert identical arrays.  10 M triangles on the GPU box's host: see profiles/r05_scene_create_10m.txt.
#include "rt_internal.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <thread>
#include <sys/mman.h>

#ifndef RT_KD_MAX_THREADS
#define RT_KD_MAX_THREADS 64          // (10 M triangles on a 256-thread host: 64 threads 5.0 s, 128 threads 5.7 s)
#endif

namespace rt {
namespace {

struct Edge {
    float t;
    uint32_t pk;                                   // kind << 31 | primitive (node-local number below the root): 0 = lower bound ("start"), 1 = upper bound ("end")
    bool operator<(const Edge &o) const { return t == o.t ? pk < o.pk : t < o.t; }      // (t, START < END, primitive)
};
inline uint32_t edge_kind(const Edge &e) { return e.pk >> 31; }
inline uint32_t edge_prim(const Edge &e) { return e.pk & 0x7fffffffu; }

struct Box { float lo[3], hi[3]; };

inline int round_to_int(double v) { return int(v + (.5 - 1.4e-11)); }          // pbrt.h:604-613
inline int log2_int(float v) {                                                    // pbrt.h:575-586
    static const float inv_log2 = 1.f / logf(2.f);
    return int(double(logf(v) * inv_log2) + (.5 - 1.4e-11));
}

// large, freshly allocated arrays on 2 MB pages where the kernel offers them (madvise mode): their first touch is a few thousand page faults instead of millions
inline void huge_pages(void *p, size_t bytes) {
    const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 4095) & ~uintptr_t(4095);
    if (bytes > (size_t(64) << 20)) madvise(reinterpret_cast<void *>(a), bytes - (a - reinterpret_cast<uintptr_t>(p)), MADV_HUGEPAGE);
}

// LIFO scratch: a node's child lists live here while its subtree is built
class Stack {
  public:
    struct Mark { size_t block, used; };
    void *alloc(size_t bytes) {
        bytes = (bytes + 63) & ~size_t(63);
        while (cur < blocks.size() && used + bytes > blocks[cur].size) { ++cur; used = 0; }
        if (cur == blocks.size()) {
            const size_t sz = std::max(bytes, std::max<size_t>(size_t(1) << 22, blocks.empty() ? 0 : blocks.back().size * 2));
            blocks.push_back(Block{std::unique_ptr<char[]>(new char[sz]), sz});
            huge_pages(blocks.back().mem.get(), sz);
            used = 0;
        }
        void *p = blocks[cur].mem.get() + used;
        used += bytes;
        return p;
    }
    template <class T> T *get(size_t n) { return static_cast<T *>(alloc(n * sizeof(T))); }
    Mark mark() const { return Mark{cur, used}; }
    void release(Mark m) { cur = m.block; used = m.used; }
  private:
    struct Block { std::unique_ptr<char[]> mem; size_t size; };
    std::vector<Block> blocks;
    size_t cur = 0, used = 0;
};

// a node's primitives (in the order the reference's classification loops produce them: the leaf order) and their bound edges, sorted, per axis
struct Lists { Edge *e[3]; int *prims; int n; };
struct To { int q0, q1; };                         // a primitive's node-local number in the below / above child, -1: not there

// sample sort with std::sort inside the buckets: the result is THE sorted sequence (total order), whatever the thread count
void parallel_sort(Edge *a, size_t n, int threads) {
    if (threads < 2 || n < (size_t(1) << 17)) { std::sort(a, a + n); return; }
    const int K = threads;
    std::vector<Edge> sample;
    const size_t ns = size_t(K) * 64;
    for (size_t i = 0; i < ns; ++i) sample.push_back(a[(n - 1) * i / (ns - 1)]);
    std::sort(sample.begin(), sample.end());
    std::vector<Edge> split;                                    // K - 1 splitters: bucket b holds split[b - 1] <= e < split[b]
    for (int b = 1; b < K; ++b) split.push_back(sample[size_t(b) * 64]);
    std::vector<size_t> count(size_t(K) * K, 0);                // [chunk][bucket]
    std::unique_ptr<Edge[]> tmp(new Edge[n]);
    huge_pages(tmp.get(), n * sizeof(Edge));
    auto bucket_of = [&](const Edge &e) { return int(std::upper_bound(split.begin(), split.end(), e) - split.begin()); };
    auto run = [&](auto fn) { ThreadGroup g; for (int t = 0; t < K; ++t) g.spawn(fn, t); g.join(); };
    run([&](int c) { const size_t lo = n * c / K, hi = n * (c + 1) / K; size_t *cnt = &count[size_t(c) * K]; for (size_t i = lo; i < hi; ++i) ++cnt[bucket_of(a[i])]; });
    std::vector<size_t> start(size_t(K) * K), bstart(K + 1, 0);
    { size_t at = 0; for (int b = 0; b < K; ++b) { bstart[b] = at; for (int c = 0; c < K; ++c) { start[size_t(c) * K + b] = at; at += count[size_t(c) * K + b]; } } bstart[K] = at; }
    run([&](int c) { const size_t lo = n * c / K, hi = n * (c + 1) / K; size_t *at = &start[size_t(c) * K]; for (size_t i = lo; i < hi; ++i) tmp[at[bucket_of(a[i])]++] = a[i]; });
    run([&](int b) { std::sort(tmp.get() + bstart[b], tmp.get() + bstart[b + 1]); std::memcpy(a + bstart[b], tmp.get() + bstart[b], (bstart[b + 1] - bstart[b]) * sizeof(Edge)); });
}

class Builder {
  public:
    Builder(const float *verts, uint32_t n, const RtAccelParams &p, KdTree &out)
        : nTris(n), prm(p), target(&out) {
        KdTree &tree = out;
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

    // ---- parallel build: the top of the tree is a fork-join (below); every subtree with <= cutoff primitives becomes a task that an
    // independent Builder turns into its own node / leaf-reference arrays; stitching them back in depth-first order reproduces the serial
    // build's arrays exactly (each subtree's construction depends only on its arguments).
    struct Task { Box nb; std::vector<int> prims; std::vector<Edge> e[3]; int depth, bad; uint32_t placeholder; KdTree sub; };
    std::vector<Task> *tasks = nullptr;
    int cutoff = 0;
    // fork-join over the top of the tree: while fork_levels > 0 a node's ABOVE child is built by a forked thread (its own Builder,
    // scratch stack, node array and task list) while this thread builds the below child in place; the forked arrays are then appended
    // with their indices rebased.  Every node's construction depends only on its own arguments, so the arrays are those of the serial build.
    int fork_levels = 0;
    int par = 1;                                   // threads this builder may use for the passes of ONE node (the top of the tree; halved at every fork)
    bool sorting_form = false;                     // every node sorts its edges itself (the reference's way, with this file's total order): the check

    void run() {
        KdTree &tree = *target;
        int depth = prm.max_depth;
        if (nTris == 0) depth = 1;                       // empty world: one empty leaf (the reference evaluates log(0) here)
        else if (depth <= 0) depth = round_to_int(8 + 1.3f * log2_int(float(nTris)));
        tree.max_depth = depth;
        sorting_form = prm.build_threads == -1;
        Box root; for (int a = 0; a < 3; ++a) { root.lo[a] = tree.bounds[a]; root.hi[a] = tree.bounds[3 + a]; }
        int threads = prm.build_threads > 0 ? prm.build_threads : int(std::thread::hardware_concurrency());
        if (threads > RT_KD_MAX_THREADS) threads = RT_KD_MAX_THREADS;
        if (sorting_form) threads = 1;
        const bool serial = threads < 2 || nTris < 20000;
        const bool log = std::getenv("PBRT_HIP_CREATE_LOG") != nullptr;
        auto tick0 = std::chrono::steady_clock::now();
        auto tick = [&](const char *what) { if (log) { auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "KDBUILD %-28s %.3f s\n", what, std::chrono::duration<double>(t - tick0).count()); tick0 = t; } };
        Lists L = root_lists(serial ? 1 : threads);
        tick("root lists (3 sorts)");
        if (serial) {
            tree.nodes.reserve(size_t(nTris) * 4 + 64);
            split(root, L, depth, 0);
            return;
        }
        int per_thread = 8;                                            // tasks per thread the cut aims at (subtree sizes vary a lot: more, smaller tasks balance the pool)
        if (const char *e = std::getenv("PBRT_HIP_KD_TASKS_PER_THREAD")) per_thread = std::max(1, std::atoi(e));
        std::vector<Task> tk; tasks = &tk; cutoff = std::max(2048, int(nTris / (size_t(per_thread) * threads)));
        fork_levels = 0; for (int t = 1; t < threads && fork_levels < 6; t <<= 1) ++fork_levels;
        par = threads;
        split(root, L, depth, 0);
        tick("top of the tree (fork-join)");
        fork_levels = 0;
        tasks = nullptr;
        stack = Stack();
        std::vector<Box>().swap(primBoxOwn);
        std::atomic<size_t> next(0);
        // largest tasks first: the pool's tail is then made of small ones
        std::vector<size_t> order(tk.size());
        for (size_t i = 0; i < tk.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return tk[x].prims.size() != tk[y].prims.size() ? tk[x].prims.size() > tk[y].prims.size() : x < y; });
        std::vector<double> idle_at(size_t(threads), 0.0);
        const auto pool_t0 = std::chrono::steady_clock::now();
        std::atomic<int> worker_id(0);
        auto worker = [&]() {
            const int me = worker_id.fetch_add(1);
            Builder sub(*this, nullptr);                           // one scratch stack per worker, reused by its tasks
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= tk.size()) { idle_at[size_t(me)] = std::chrono::duration<double>(std::chrono::steady_clock::now() - pool_t0).count(); return; }
                Task &t = tk[order[k]];
                sub.target = &t.sub;
                sub.build_task(t);
            }
        };
        { ThreadGroup pool; for (int t = 0; t < threads; ++t) pool.spawn(worker); pool.join(); }
        tick("task pool");
        if (log) {
            std::sort(idle_at.begin(), idle_at.end());
            std::fprintf(stderr, "KDBUILD %zu tasks, cutoff %d, %d threads; workers ran out of tasks after %.3f (first) / %.3f (median) / %.3f s (last)\n", tk.size(), cutoff, threads,
                         idle_at.front(), idle_at[idle_at.size() / 2], idle_at.back());
        }
        stitch(tk, threads);
        tick("stitch");
    }

    // subtree builder borrowing the parent's primitive boxes and parameters
    Builder(const Builder &parent, KdTree *out) : nTris(parent.nTris), prm(parent.prm), target(out), primBoxPtr(parent.primBoxPtr) { sorting_form = parent.sorting_form; }

  private:
    uint32_t nTris;
    RtAccelParams prm;
    KdTree *target;                                 // where this builder's nodes and leaf references go
    std::vector<Box> primBoxOwn;
    const std::vector<Box> *primBoxPtr = nullptr;
    Stack stack;
    KdTree &out() { return *target; }

    void build_task(Task &t) {
        const Stack::Mark m = stack.mark();
        Lists L; L.n = int(t.prims.size());
        L.prims = t.prims.data();                                        // the task's own arrays are this node's lists (read only)
        for (int a = 0; a < 3; ++a) L.e[a] = sorting_form ? stack.get<Edge>(size_t(2) * std::max(L.n, 1)) : t.e[a].data();
        split(t.nb, L, t.depth, t.bad);
        std::vector<int>().swap(t.prims);
        for (int a = 0; a < 3; ++a) std::vector<Edge>().swap(t.e[a]);
        stack.release(m);
    }

    // the root's lists: every primitive, its bounds on each axis, sorted (three axes side by side, each with a share of the threads)
    Lists root_lists(int threads) {
        Lists L; L.n = int(nTris);
        L.prims = stack.get<int>(size_t(std::max(L.n, 1)));
        for (uint32_t i = 0; i < nTris; ++i) L.prims[i] = int(i);
        for (int a = 0; a < 3; ++a) L.e[a] = stack.get<Edge>(size_t(2) * std::max(L.n, 1));
        if (sorting_form) return L;
        auto one = [&](int a, int th) {
            Edge *e = L.e[a];
            const std::vector<Box> &pb = *primBoxPtr;
            for (uint32_t i = 0; i < nTris; ++i) { e[2 * size_t(i)] = Edge{pb[i].lo[a], i}; e[2 * size_t(i) + 1] = Edge{pb[i].hi[a], i | 0x80000000u}; }
            parallel_sort(e, size_t(2) * nTris, th);
        };
        if (threads < 3) { for (int a = 0; a < 3; ++a) one(a, threads); }
        else {
            ThreadGroup g; g.spawn(one, 1, threads / 3); g.spawn(one, 2, threads / 3);
            one(0, threads - 2 * (threads / 3));
            g.join();
        }
        return L;
    }

    void leaf(uint32_t at, const int *prims, int n) {
        KdTree &tr = out();
        tr.nodes[at].x = (uint32_t(n) << 2) | 3u;
        if (n == 0) tr.nodes[at].y = 0;
        else if (n == 1) tr.nodes[at].y = uint32_t(prims[0]);
        else {
            tr.nodes[at].y = uint32_t(tr.leaf_refs.size());
            const size_t at0 = tr.leaf_refs.size();
            tr.leaf_refs.resize(at0 + size_t(n));                           // (no-init vector: one bounds check, one copy)
            std::memcpy(tr.leaf_refs.data() + at0, prims, size_t(n) * sizeof(uint32_t));
        }
    }

    // L: this node's lists (in the caller's scratch; the caller releases them)
    void split(const Box &nb, const Lists &L, int depth, int bad) {
        KdTree &tr = out();
        const int n = L.n;
        const int *prims = L.prims;
        const uint32_t me = uint32_t(tr.nodes.size());
        tr.nodes.push_back(Node{0, 0});
        if (n <= prm.max_prims || depth == 0) { leaf(me, prims, n); return; }
        if (tasks && n <= cutoff) {                       // hand this subtree to the task pool
            Task t; t.nb = nb; t.prims.assign(prims, prims + n); t.depth = depth; t.bad = bad; t.placeholder = me;
            if (!sorting_form) for (int a = 0; a < 3; ++a) t.e[a].assign(L.e[a], L.e[a] + size_t(2) * n);
            tr.nodes[me].x = 0xFFFFFFFFu; tr.nodes[me].y = uint32_t(tasks->size());
            tasks->push_back(std::move(t));
            return;
        }

        int bestAxis = -1, bestEdge = -1, bestBelow = 0, bestAbove = 0;
        float bestCost = INFINITY;
        const float leafCost = prm.isect_cost * float(n);
        const float d[3] = {nb.hi[0] - nb.lo[0], nb.hi[1] - nb.lo[1], nb.hi[2] - nb.lo[2]};
        const float totalSA = (2.f * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]));
        const float invTotalSA = 1.f / totalSA;
        int axis = (d[0] > d[1] && d[0] > d[2]) ? 0 : ((d[1] > d[2]) ? 1 : 2);

        for (int attempt = 0;; ++attempt) {
            Edge *e = L.e[axis];
            if (sorting_form) {                                            // the reference's way (kdtree.cpp:236-246): build and sort this node's edges
                for (int i = 0; i < n; ++i) {                              // (node-local primitive numbers follow the primitive's number: same order)
                    const int p = prims[i];
                    e[2 * i] = Edge{(*primBoxPtr)[p].lo[axis], uint32_t(p)};
                    e[2 * i + 1] = Edge{(*primBoxPtr)[p].hi[axis], uint32_t(p) | 0x80000000u};
                }
                std::sort(e, e + 2 * n);
            }
            int nBelow = 0, nAbove = n;
            const int o0 = axis == 0 ? 1 : 0;                              // {1,2},{0,2},{0,1}
            const int o1 = axis == 2 ? 1 : 2;
            for (int i = 0; i < 2 * n; ++i) {
                const uint32_t kind = edge_kind(e[i]);
                if (kind == 1) --nAbove;
                const float t = e[i].t;
                if (t > nb.lo[axis] && t < nb.hi[axis]) {
                    const float belowSA = 2 * (d[o0] * d[o1] + (t - nb.lo[axis]) * (d[o0] + d[o1]));
                    const float aboveSA = 2 * (d[o0] * d[o1] + (nb.hi[axis] - t) * (d[o0] + d[o1]));
                    const float pB = belowSA * invTotalSA, pA = aboveSA * invTotalSA;
                    const float bonus = (nAbove == 0 || nBelow == 0) ? prm.empty_bonus : 0.f;
                    const float cost = prm.trav_cost + prm.isect_cost * (1.f - bonus) * (pB * nBelow + pA * nAbove);
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestEdge = i; bestBelow = nBelow; bestAbove = nAbove; }
                }
                if (kind == 0) ++nBelow;
            }
            if (bestAxis == -1 && attempt < 2) { axis = (axis + 1) % 3; continue; }
            break;
        }
        if (bestCost > leafCost) ++bad;
        if ((bestCost > 4.f * leafCost && n < 16) || bestAxis == -1 || bad == 3) { leaf(me, prims, n); return; }

        // classification (kdtree.cpp:284-289): below = the primitives whose lower bound comes before the split edge, above = those whose upper bound
        // comes after it, each in edge order (the order of a leaf's primitives); bestBelow / bestAbove are exactly their numbers
        const Edge *e = L.e[bestAxis];
        const int n0 = bestBelow, n1 = bestAbove;
        const float ts = e[bestEdge].t;
        uint32_t bits; std::memcpy(&bits, &ts, 4);
        tr.nodes[me].x = (bits & ~3u) | uint32_t(bestAxis);
        Box b0 = nb, b1 = nb;
        b0.hi[bestAxis] = b1.lo[bestAxis] = ts;
        const bool fork = tasks && fork_levels > 0 && n1 > cutoff && n0 > cutoff;
        std::unique_ptr<KdTree> fsub; std::unique_ptr<std::vector<Task>> fTasks; std::unique_ptr<Builder> fb;
        if (fork) { fsub.reset(new KdTree); fTasks.reset(new std::vector<Task>); fb.reset(new Builder(*this, fsub.get())); }
        const Stack::Mark m1 = stack.mark();
        Stack &s1 = fork ? fb->stack : stack;                              // the above child's lists live where the above child is built
        Lists c1; c1.n = n1; c1.prims = s1.get<int>(size_t(n1) + 1);
        for (int a = 0; a < 3; ++a) c1.e[a] = s1.get<Edge>(size_t(2) * n1 + 1);
        const Stack::Mark m0 = stack.mark();
        Lists c0; c0.n = n0; c0.prims = stack.get<int>(size_t(n0) + 1);
        for (int a = 0; a < 3; ++a) c0.e[a] = stack.get<Edge>(size_t(2) * n0 + 1);
        if (sorting_form) {
            int k0 = 0, k1 = 0;
            for (int i = 0; i < bestEdge; ++i) if (edge_kind(e[i]) == 0) c0.prims[k0++] = int(edge_prim(e[i]));
            for (int i = bestEdge + 1; i < 2 * n; ++i) if (edge_kind(e[i]) == 1) c1.prims[k1++] = int(edge_prim(e[i]));
        } else {
            const Stack::Mark ms = stack.mark();
            // node-local number of each primitive in the children (-1: not there); one record per primitive: the filter below looks it up once per edge,
            // in the order of the sorted lists, i.e. at random (at the root of a 10 M-triangle build that look-up IS the cost)
            To *to = stack.get<To>(size_t(n));
            const int P = (tasks && par > 1 && n >= (1 << 16)) ? std::min(par, 64) : 1;      // the top of the tree: this node's passes cut into chunks, one thread each
            if (P == 1) {
                // (branch-free: whether an edge is a START / goes to a child is a coin toss to the branch predictor; every store below is unconditional, into the
                // real slot or a dummy one, and the cursors advance by 0 or 1.  The children's arrays have one spare element for the last store.)
                std::memset(to, 0xff, size_t(n) * sizeof(To));
                int k0 = 0, k1 = 0, dummy = 0;
                for (int i = 0; i < bestEdge; ++i) {
                    const uint32_t p = edge_prim(e[i]); const bool st = edge_kind(e[i]) == 0;
                    (st ? to[p].q0 : dummy) = k0; c0.prims[k0] = prims[p]; k0 += st;
                }
                for (int i = bestEdge + 1; i < 2 * n; ++i) {
                    const uint32_t p = edge_prim(e[i]); const bool en = edge_kind(e[i]) == 1;
                    (en ? to[p].q1 : dummy) = k1; c1.prims[k1] = prims[p]; k1 += en;
                }
                for (int a = 0; a < 3; ++a) {                              // the children's lists: this node's, filtered
                    const Edge *src = L.e[a];
                    Edge *d0 = c0.e[a], *d1 = c1.e[a];
                    for (int i = 0; i < 2 * n; ++i) {
                        const Edge x = src[i];
                        const uint32_t kb = x.pk & 0x80000000u;
                        const To q = to[x.pk & 0x7fffffffu];
                        *d0 = Edge{x.t, kb | uint32_t(q.q0)}; d0 += q.q0 >= 0;
                        *d1 = Edge{x.t, kb | uint32_t(q.q1)}; d1 += q.q1 >= 0;
                    }
                }
            } else {
                // the same passes, chunked: every chunk first counts what it will write, a prefix sum gives it its place, then it writes -- the output is
                // the sequential pass's, whatever P is
                const size_t N2 = size_t(2) * n;
                auto chunk = [&](size_t total, int c) { return std::make_pair(total * size_t(c) / size_t(P), total * size_t(c + 1) / size_t(P)); };
                auto run = [&](auto fn) { ThreadGroup g; for (int c = 1; c < P; ++c) g.spawn(fn, c); fn(0); g.join(); };
                std::vector<size_t> cnt0(size_t(P) * 4, 0);                 // per chunk: [classification: starts, ends] then reused per axis
                run([&](int c) {
                    const auto r = chunk(size_t(n), c);
                    std::memset(to + r.first, 0xff, (r.second - r.first) * sizeof(To));
                    const auto q = chunk(N2, c);
                    size_t s0 = 0, s1 = 0;
                    for (size_t i = q.first; i < q.second; ++i) {
                        if (i < size_t(bestEdge)) s0 += edge_kind(e[i]) == 0;
                        else if (i > size_t(bestEdge)) s1 += edge_kind(e[i]) == 1;
                    }
                    cnt0[size_t(c) * 4] = s0; cnt0[size_t(c) * 4 + 1] = s1;
                });
                { size_t a0 = 0, a1 = 0; for (int c = 0; c < P; ++c) { const size_t s0 = cnt0[size_t(c) * 4], s1 = cnt0[size_t(c) * 4 + 1]; cnt0[size_t(c) * 4] = a0; cnt0[size_t(c) * 4 + 1] = a1; a0 += s0; a1 += s1; } }
                run([&](int c) {
                    const auto q = chunk(N2, c);
                    int k0 = int(cnt0[size_t(c) * 4]), k1 = int(cnt0[size_t(c) * 4 + 1]);
                    for (size_t i = q.first; i < q.second; ++i) {
                        if (i < size_t(bestEdge)) { if (edge_kind(e[i]) == 0) { const uint32_t p = edge_prim(e[i]); to[p].q0 = k0; c0.prims[k0++] = prims[p]; } }
                        else if (i > size_t(bestEdge)) { if (edge_kind(e[i]) == 1) { const uint32_t p = edge_prim(e[i]); to[p].q1 = k1; c1.prims[k1++] = prims[p]; } }
                    }
                });
                std::vector<size_t> cnt(size_t(P) * 6, 0);                  // [chunk][axis][child]
                run([&](int c) {
                    const auto q = chunk(N2, c);
                    for (int a = 0; a < 3; ++a) {
                        const Edge *src = L.e[a];
                        size_t s0 = 0, s1 = 0;
                        for (size_t i = q.first; i < q.second; ++i) { const To t = to[src[i].pk & 0x7fffffffu]; s0 += t.q0 >= 0; s1 += t.q1 >= 0; }
                        cnt[size_t(c) * 6 + 2 * a] = s0; cnt[size_t(c) * 6 + 2 * a + 1] = s1;
                    }
                });
                for (int k = 0; k < 6; ++k) { size_t at = 0; for (int c = 0; c < P; ++c) { const size_t v = cnt[size_t(c) * 6 + k]; cnt[size_t(c) * 6 + k] = at; at += v; } }
                run([&](int c) {
                    const auto q = chunk(N2, c);
                    for (int a = 0; a < 3; ++a) {
                        const Edge *src = L.e[a];
                        Edge *d0 = c0.e[a] + cnt[size_t(c) * 6 + 2 * a], *d1 = c1.e[a] + cnt[size_t(c) * 6 + 2 * a + 1];
                        for (size_t i = q.first; i < q.second; ++i) {
                            const Edge x = src[i];
                            const uint32_t kb = x.pk & 0x80000000u;
                            const To t = to[x.pk & 0x7fffffffu];
                            if (t.q0 >= 0) *d0++ = Edge{x.t, kb | uint32_t(t.q0)};
                            if (t.q1 >= 0) *d1++ = Edge{x.t, kb | uint32_t(t.q1)};
                        }
                    }
                });
            }
            stack.release(ms);
        }
        if (fork) {
            // above child in a forked thread (own scratch, arrays and task list), below child here; then append the forked arrays behind ours
            const int levels = fork_levels - 1, cut = cutoff;
            const int par_all = par;
            ThreadGroup th;
            th.spawn([&, levels, cut, par_all]() {
                fb->tasks = fTasks.get(); fb->cutoff = cut; fb->fork_levels = levels; fb->par = std::max(1, par_all / 2);
                fb->split(b1, c1, depth - 1, bad);
            });
            const int saved = fork_levels; fork_levels = levels; par = std::max(1, par_all - par_all / 2);
            split(b0, c0, depth - 1, bad);
            fork_levels = saved; par = par_all;
            th.join();
            stack.release(m1);
            const KdTree &sub = *fsub;
            const uint32_t nodeBase = uint32_t(tr.nodes.size()), refBase = uint32_t(tr.leaf_refs.size()), taskBase = uint32_t(tasks->size());
            tr.nodes[me].y = nodeBase;
            for (const Node &sn : sub.nodes) {
                Node o = sn;
                if (sn.x == 0xFFFFFFFFu) o.y = sn.y + taskBase;              // placeholder of a pooled subtree
                else if ((sn.x & 3u) != 3u) o.y = sn.y + nodeBase;           // interior: above-child index
                else if ((sn.x >> 2) > 1) o.y = sn.y + refBase;              // leaf of the top with a reference list
                tr.nodes.push_back(o);
            }
            tr.leaf_refs.insert(tr.leaf_refs.end(), sub.leaf_refs.begin(), sub.leaf_refs.end());
            for (Task &t : *fTasks) { t.placeholder += nodeBase; tasks->push_back(std::move(t)); }
            return;
        }
        split(b0, c0, depth - 1, bad);
        stack.release(m0);
        out().nodes[me].y = uint32_t(out().nodes.size());
        split(b1, c1, depth - 1, bad);
        stack.release(m1);
    }

    // stitch: the top's nodes in depth-first order with every placeholder replaced by its task's subtree (indices rebased).  Where each piece
    // lands follows from the sizes alone (a serial walk over the small top); the pieces are then copied by all threads.
    void stitch(std::vector<Task> &tk, int threads) {
        KdTree &tree = *target;
        NodeVec top; top.swap(tree.nodes);
        RefVec topRefs; topRefs.swap(tree.leaf_refs);
        struct Piece { uint32_t task, nodeBase, refBase; };
        std::vector<Piece> pieces; pieces.reserve(tk.size());
        // pass 1: final position of every top node and task
        std::vector<uint32_t> pos(top.size());
        size_t nodeAt = 0, refAt = 0;
        struct Walk {
            const NodeVec &top; std::vector<Task> &tk; std::vector<uint32_t> &pos; std::vector<Piece> &pieces; size_t &nodeAt, &refAt;
            std::vector<uint32_t> refPos;
            void go(uint32_t i) {
                const Node nd = top[i];
                if (nd.x == 0xFFFFFFFFu) {
                    const Task &t = tk[nd.y];
                    pos[i] = uint32_t(nodeAt);
                    pieces.push_back(Piece{nd.y, uint32_t(nodeAt), uint32_t(refAt)});
                    nodeAt += t.sub.nodes.size(); refAt += t.sub.leaf_refs.size();
                    return;
                }
                pos[i] = uint32_t(nodeAt++);
                if ((nd.x & 3u) == 3u) { refPos[i] = uint32_t(refAt); if ((nd.x >> 2) > 1) refAt += nd.x >> 2; return; }
                go(i + 1);
                go(nd.y);
            }
        } walk{top, tk, pos, pieces, nodeAt, refAt, std::vector<uint32_t>(top.size(), 0)};
        walk.go(0);
        const bool slog = std::getenv("PBRT_HIP_CREATE_LOG") != nullptr;
        auto st0 = std::chrono::steady_clock::now();
        auto stick = [&](const char *what) { if (slog) { auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "KDBUILD   stitch: %-22s %.3f s\n", what, std::chrono::duration<double>(t - st0).count()); st0 = t; } };
        tree.nodes.resize(nodeAt); tree.leaf_refs.resize(refAt);
        // the final arrays on 2 MB pages where the kernel offers them: their first touch by the copying threads is a few thousand page faults instead of 1.5 M
        huge_pages(tree.nodes.data(), tree.nodes.size() * sizeof(Node)); huge_pages(tree.leaf_refs.data(), tree.leaf_refs.size() * sizeof(uint32_t));
        stick("walk + resize");
        // pass 2: the top's own nodes
        for (size_t i = 0; i < top.size(); ++i) {
            const Node nd = top[i];
            if (nd.x == 0xFFFFFFFFu) continue;
            Node o = nd;
            if ((nd.x & 3u) == 3u) {
                const uint32_t np = nd.x >> 2;
                if (np > 1) { o.y = walk.refPos[i]; for (uint32_t k = 0; k < np; ++k) tree.leaf_refs[walk.refPos[i] + k] = topRefs[nd.y + k]; }
            } else o.y = pos[nd.y];
            tree.nodes[pos[i]] = o;
        }
        // pass 3: the tasks' subtrees
        std::atomic<size_t> next(0);
        auto worker = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= pieces.size()) return;
                const Piece &pc = pieces[k];
                Task &t = tk[pc.task];
                Node *dst = tree.nodes.data() + pc.nodeBase;
                const size_t nn = t.sub.nodes.size();
                for (size_t j = 0; j < nn; ++j) {
                    Node o = t.sub.nodes[j];
                    if ((o.x & 3u) != 3u) o.y += pc.nodeBase;              // interior: above-child index
                    else if ((o.x >> 2) > 1) o.y += pc.refBase;            // leaf with a reference list
                    dst[j] = o;
                }
                if (!t.sub.leaf_refs.empty()) std::memcpy(tree.leaf_refs.data() + pc.refBase, t.sub.leaf_refs.data(), t.sub.leaf_refs.size() * sizeof(uint32_t));
                NodeVec().swap(t.sub.nodes); RefVec().swap(t.sub.leaf_refs);
            }
        };
        { ThreadGroup pool; for (int t = 0; t < threads; ++t) pool.spawn(worker); pool.join(); }
        stick("copy + free");
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
