// leaf_layout.cpp -- host side of the flat traversal's leaves: one record per primitive and the entries that walk a leaf's primitives (rt_leaf_entries.h).
// Plain C++ (no device code): rt_scene_create calls it, and rt_accel_leaf_layout exposes it to the CPU tests (tests/test_host.py).
#include "rt_internal.h"
#include "rt_leaf_entries.h"
#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>

namespace rt {

// One record per primitive for the flat traversal (rt_device.h DevScene::ltris / lrefs / tnodes; the entry encoding: rt_traverse.h RT_LE_*).
// Rounds 2-5 kept one 48-byte copy per leaf REFERENCE, a leaf's copies side by side: 25.1 M copies of the benchmark soup's 1 M triangles (1.2 GB;
// 12 GB at 10 M triangles) that no cache level holds.  Now a primitive has ONE record, placed where the depth-first leaf walk first meets it (so the
// primitives of neighbouring leaves are neighbours), RT_TRI_STRIDE float4 units apart; a leaf node names its first primitive inline, a leaf of two
// the second one in its word 1, a larger leaf the index of its remaining entries in `lrefs` (the reference's own form, kdtree.cpp:55-64).
// `copies` (PBRT_HIP_LEAF_COPIES, measurements only): every reference gets a record of its own again -- the same kernel, the old footprint.
// `runs`: the leaves own runs of consecutive records and word 1 is the primitive count (DevScene::leaf_runs: rounds 2-5's layout, without the line
// alignment; what scenes of a few thousand references use -- cache resident, bound by instruction issue, where fetching entries costs 3 %).
bool leaf_cursor_layout(const NodeVec &nodes, const RefVec &leaf_refs, uint32_t n_tris, bool copies, bool runs, LeafLayout &o) {
    copies = copies || runs;
    const size_t N = nodes.size();
    o.tnodes.resize(N);
    const size_t B = size_t(1) << 18, nb = (N + B - 1) / B;
    const size_t nthreads = nb < 4 ? 1 : std::min<size_t>(nb, std::max(1u, std::min(64u, std::thread::hardware_concurrency())));
    auto run = [&](auto fn) {
        if (nthreads == 1) { for (size_t b = 0; b < nb; ++b) fn(b); return; }
        std::atomic<size_t> next(0);
        ThreadGroup pool;
        for (size_t t = 0; t < nthreads; ++t) pool.spawn([&] { for (;;) { const size_t b = next.fetch_add(1); if (b >= nb) return; fn(b); } });
    };
    auto leaf_n = [&](const Node &n) -> uint32_t { return (n.x & 3u) == 3u ? n.x >> 2 : 0u; };
    auto ref = [&](const Node &n, uint32_t np, uint32_t k) -> uint32_t { return np == 1 ? n.y : leaf_refs[n.y + k]; };
    // pass 1: where the walk first meets every primitive (64-bit key = node << 32 | position in the leaf; minimum over its references)
    std::unique_ptr<std::atomic<uint64_t>[]> first;
    if (!copies) {
        first.reset(new std::atomic<uint64_t>[size_t(n_tris) + 1]);
        for (size_t i = 0; i <= n_tris; ++i) first[i].store(~0ull, std::memory_order_relaxed);
        run([&](size_t b) {
            const size_t hi = std::min(N, (b + 1) * B);
            for (size_t i = b * B; i < hi; ++i) {
                const Node n = nodes[i]; const uint32_t np = leaf_n(n);
                for (uint32_t k = 0; k < np; ++k) {
                    const uint64_t key = uint64_t(i) << 32 | k;
                    std::atomic<uint64_t> &f = first[ref(n, np, k)];
                    uint64_t cur = f.load(std::memory_order_relaxed);
                    while (key < cur && !f.compare_exchange_weak(cur, key, std::memory_order_relaxed)) {}
                }
            }
        });
    }
    // pass 2: per block of nodes, the records it opens and the list entries its leaves of three or more need
    std::vector<size_t> slots(nb + 1, 0), lists(nb + 1, 0);
    run([&](size_t b) {
        const size_t hi = std::min(N, (b + 1) * B);
        size_t ns = 0, nl = 0;
        for (size_t i = b * B; i < hi; ++i) {
            const Node n = nodes[i]; const uint32_t np = leaf_n(n);
            if (np >= 3 && !runs) nl += (np - 1 + 1) & ~size_t(1);            // lists start at even indices (the cursor is stored halved)
            if (copies) ns += np;
            else for (uint32_t k = 0; k < np; ++k) ns += first[ref(n, np, k)].load(std::memory_order_relaxed) == (uint64_t(i) << 32 | k);
        }
        slots[b + 1] = ns; lists[b + 1] = nl;
    });
    for (size_t b = 0; b < nb; ++b) { slots[b + 1] += slots[b]; lists[b + 1] += lists[b]; }
    o.n_slots = slots[nb];
    if (o.n_slots * RT_TRI_STRIDE >= RT_LE_POS || lists[nb] / 2 >= RT_LE_POS) return false;
    o.slot_prim.resize(o.n_slots);
    o.lrefs.resize(lists[nb] ? lists[nb] : 1);
    // pass 3: a primitive's slot (the record it shares, or one per reference)
    std::vector<uint32_t> slot_of;
    if (!copies) {
        slot_of.assign(size_t(n_tris) + 1, 0u);
        run([&](size_t b) {
            const size_t hi = std::min(N, (b + 1) * B);
            size_t at = slots[b];
            for (size_t i = b * B; i < hi; ++i) {
                const Node n = nodes[i]; const uint32_t np = leaf_n(n);
                for (uint32_t k = 0; k < np; ++k) {
                    const uint32_t p = ref(n, np, k);
                    if (first[p].load(std::memory_order_relaxed) == (uint64_t(i) << 32 | k)) { slot_of[p] = uint32_t(at); o.slot_prim[at++] = p; }
                }
            }
        });
    }
    // pass 4: the leaves in entry form
    run([&](size_t b) {
        const size_t hi = std::min(N, (b + 1) * B);
        size_t at = slots[b], lat = lists[b];
        for (size_t i = b * B; i < hi; ++i) {
            const Node n = nodes[i];
            o.tnodes[i] = n;
            if ((n.x & 3u) != 3u) continue;
            const uint32_t np = n.x >> 2;
            if (np == 0) { o.tnodes[i].x = RT_LE_NONE; o.tnodes[i].y = ~RT_LE_POS; continue; }       // entry RT_LE_NONE (runs: + a count that is never read)
            auto pos = [&](uint32_t k) -> uint32_t {
                if (copies) { o.slot_prim[at + k] = ref(n, np, k); return uint32_t(at + k) * RT_TRI_STRIDE; }
                return slot_of[ref(n, np, k)] * RT_TRI_STRIDE;
            };
            o.tnodes[i].x = pos(0) << 2 | 3u;
            if (runs) { for (uint32_t k = 1; k < np; ++k) pos(k); o.tnodes[i].y = (np > 1 ? RT_LE_MORE : 0u) | np; }
            else if (np == 1) o.tnodes[i].y = 0u;
            else if (np == 2) o.tnodes[i].y = RT_LE_MORE | pos(1);
            else {
                o.tnodes[i].y = RT_LE_MORE | RT_LE_LIST | uint32_t(lat / 2);
                for (uint32_t k = 1; k < np; ++k) o.lrefs[lat++] = pos(k) | (k + 1 < np ? RT_LE_MORE | RT_LE_LIST : 0u);
                if (lat & 1) o.lrefs[lat++] = RT_LE_NONE;                          // padding, never read
            }
            if (copies) at += np;
        }
    });
    if (lists[nb] == 0) o.lrefs[0] = 0u;
    return true;
}

}  // namespace rt
