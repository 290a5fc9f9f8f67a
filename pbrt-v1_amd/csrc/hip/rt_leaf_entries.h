// rt_leaf_entries.h -- the encoding of a kd leaf's primitives in the flat traversal: shared by the kernels (rt_traverse.h) and the host-side layout (leaf_layout.cpp)
#pragma once
// ---- the leaf cursor of the flat traversal (round 6: ONE record per primitive) ------------------------------------------------------------
// DevScene::ltris holds one record per distinct primitive, in the order the depth-first leaf walk first meets them (RT_TRI_STRIDE float4 units
// apart), instead of one 48-byte copy per leaf REFERENCE (25.1 M copies of 1 M triangles = 1.2 GB on the benchmark soup, 12 GB at 10 M triangles;
// the reference itself keeps indices, kdtree.cpp:55-64).  A leaf's primitives are walked through "entries" = position (30 bits) | flags:
//   RT_LE_MORE   another primitive follows this one,
//   RT_LE_LIST   ... and its entry is read from DevScene::lrefs at the cursor (otherwise the cursor itself is that entry: leaves of two).
// A leaf node carries its FIRST entry inline -- word 0 = position << 2 | 3, the entry's two flags in the top bits of word 1 -- and in the low
// 30 bits of word 1 the cursor: the second entry (leaves of two: 72 % of the soup's leaf references sit in leaves of one or two and need no
// index fetch at all) or HALF the index of the leaf's remaining entries in `lrefs` (lists start at even indices: 2^31 entries, the 10 M-triangle
// soup has 0.9 G), which are requested TOGETHER with the record of the primitive before them, so no test waits for two dependent round trips.
// An empty leaf has the entry RT_LE_NONE.  The counting twins keep "the leaf has more than one primitive" (= RT_LE_MORE of its first entry: the
// reference walks a list there, the leaf_refs counter) in Trav::li.
// DevScene::leaf_runs (scenes of a few thousand references: everything is cache resident and the kernel is bound by instruction issue): every leaf owns a
// run of consecutive records instead -- word 1 = RT_LE_MORE (if more than one) | the number of primitives, the cursor counts down, no entry is ever fetched.
#ifndef RT_TRI_STRIDE
#define RT_TRI_STRIDE 4           // float4 units between two records: 4 = a 48-byte record never straddles two 64-byte sectors / 128-byte lines
#endif
#define RT_LE_MORE 0x80000000u
#define RT_LE_LIST 0x40000000u
#define RT_LE_POS 0x3fffffffu
#define RT_LE_NONE 0xffffffffu
