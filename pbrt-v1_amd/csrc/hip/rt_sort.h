// rt_sort.h -- device-wide sort of (key, queue position) pairs for the pipeline's ray queue (rt_sort.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace rt {
size_t sort_pairs_temp_bytes(size_t n);
hipError_t sort_pairs(void *temp, size_t temp_bytes, const unsigned *keys_in, unsigned *keys_out, const unsigned *vals_in, unsigned *vals_out,
                      size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t stream);
}  // namespace rt
