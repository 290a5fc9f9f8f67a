// rt_top_fetch.h -- the two pair records of a two-level step (kdp_step, rt_traverse.h) fetched from LDS by the lanes whose node lies in the
// workgroup's copy of the tree's top levels and from HBM by the others, in ONE instruction group.
//
// Written as one asm block because of what the compiler makes of the plain form (`if (in_top) A = lds[..]; else A = global[..];`): both branches
// write the same registers, the lanes are disjoint (the branches run under complementary exec masks), but the register allocator sees a
// write-after-write on A and puts `s_waitcnt vmcnt(0)` between the global load and the LDS read -- the whole HBM round trip in front of the LDS
// read.  Here the four loads are issued back to back under their lane masks and waited for once; the wait sits inside the block, so nothing the
// compiler schedules around it can read a register that is still in flight.
#pragma once
#include "rt_device.h"

namespace rt {

typedef unsigned rt_u32x4 __attribute__((ext_vector_type(4)));

// m_l1 / m_l2: lanes that read record 1 / record 2 from LDS (byte addresses la, lb); m_g1 / m_g2: lanes that read them from HBM (ga, gb).
// Lanes in none of the masks get undefined words (the step commits its results under lane masks).
RT_DEV void pair_fetch_mixed(uint4 &A, uint4 &B, unsigned long long m_l1, unsigned long long m_l2, unsigned long long m_g1, unsigned long long m_g2,
                             unsigned la, unsigned lb, const uint4 RT_G *ga, const uint4 RT_G *gb) {
#if defined(__HIP_DEVICE_COMPILE__)
    rt_u32x4 a, b;
    unsigned long long save;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_and_b64 exec, %[save], %[ml1]\n\t"
        "ds_read_b128 %[a], %[la]\n\t"
        "s_and_b64 exec, %[save], %[ml2]\n\t"
        "ds_read_b128 %[b], %[lb]\n\t"
        "s_and_b64 exec, %[save], %[mg1]\n\t"
        "global_load_dwordx4 %[a], %[ga], off\n\t"
        "s_and_b64 exec, %[save], %[mg2]\n\t"
        "global_load_dwordx4 %[b], %[gb], off\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        : [a] "=&v"(a), [b] "=&v"(b), [save] "=&s"(save)
        : [ml1] "s"(m_l1), [ml2] "s"(m_l2), [mg1] "s"(m_g1), [mg2] "s"(m_g2), [la] "v"(la), [lb] "v"(lb), [ga] "v"(ga), [gb] "v"(gb)
        : "memory", "scc");
    A = make_uint4(a.x, a.y, a.z, a.w);
    B = make_uint4(b.x, b.y, b.z, b.w);
#endif
}

}  // namespace rt
