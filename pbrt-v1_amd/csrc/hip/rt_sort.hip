// rt_sort.hip -- ordering the ray queue of the pipeline (rt_pipeline.h) by where the rays START in the tree.
//
// Why: at 1 M triangles a ray visits ~80 kd nodes, ~30 of them on the way down from the root to the leaf that holds its origin; the
// rays of a queue in slot order start all over the scene, so the 64 lanes of a trace wave ask for 64 different lines per step and
// the kernel runs at the chip's gather rate.  Rays sorted by the Morton code of their entry point share the
// descent: one line request serves the wave.
//
// This translation unit only wraps the device-wide LSD radix sort of (key, queue position) pairs (rocPRIM, the library primitive for
// a plain sort; the keys are produced by pipe_shade_kernel and consumed by pipe_trace_kernel).  Results of a frame never depend on the
// order (each ray's hit is a function of the ray alone; counters are sums).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "rt_sort.h"

namespace rt {

size_t sort_pairs_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned *)nullptr, (unsigned *)nullptr, (const unsigned *)nullptr, (unsigned *)nullptr,
                                    n, 0u, 32u, (hipStream_t) nullptr, false);
    return bytes;
}

hipError_t sort_pairs(void *temp, size_t temp_bytes, const unsigned *keys_in, unsigned *keys_out, const unsigned *vals_in, unsigned *vals_out,
                      size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream, false);
}

}  // namespace rt
