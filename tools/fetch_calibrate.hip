// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access patterns (MI355X_MICROARCH.md,
// HBM section: only wide coalesced streaming reads are calibrated there -- "calibrate on a known byte count in your own
// access pattern").  Each kernel moves a known number of bytes / touches a known number of 64-byte sectors of a buffer
// much larger than L2 + Infinity Cache (2 GiB), once.
//   stream16   : coalesced 16 B per lane                        (the guide's calibrated case: expect FETCH_SIZE = 1/2)
//   gather8    : one random 8-byte node per lane                (kd-tree node fetch)
//   gather48   : one random 48-byte record per lane, 3 x 16 B   (triangle fetch)
//   write32    : coalesced 32-byte records, 2 x 16 B per lane   (sample-record writes)
// build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calibrate.hip -o /tmp/fetch_calibrate
// run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o t -- /tmp/fetch_calibrate   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
__global__ void stream16(const float4 *p, size_t n, float *out) {
    float acc = 0.f;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *out = acc;
}
__global__ void gather8(const uint2 *p, uint32_t mask, uint32_t per_thread, float *out) {
    uint32_t acc = 0, h = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_thread; ++k) { h = pcg(h + k * 0x9E3779B9u); uint2 v = p[h & mask]; acc += v.x ^ v.y; }
    if (acc == 0x12345678u) *out = 1.f;
}
struct Rec48 { float4 a, b, c; };
__global__ void gather48(const Rec48 *p, uint32_t mask, uint32_t per_thread, float *out) {
    float acc = 0.f; uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_thread; ++k) { h = pcg(h + k * 0x9E3779B9u); const Rec48 &r = p[h & mask]; acc += r.a.x + r.b.y + r.c.z; }
    if (acc == 123.456f) *out = acc;
}
__global__ void write32(float4 *p, size_t nrec) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nrec; i += size_t(gridDim.x) * blockDim.x) {
        p[2 * i] = make_float4(float(i), 1.f, 2.f, 3.f); p[2 * i + 1] = make_float4(4.f, 5.f, 6.f, 7.f);
    }
}
int main() {
    const size_t bytes = size_t(2) << 30;
    void *buf; float *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&out, 4) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 8, block = 256;
    const uint32_t per_thread = 256;                                    // 2048*256*256 = 134,217,728 gathers
    hipLaunchKernelGGL(stream16, dim3(grid), dim3(block), 0, 0, (const float4 *)buf, bytes / 16, out);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(gather8, dim3(grid), dim3(block), 0, 0, (const uint2 *)buf, uint32_t(bytes / 8 - 1), per_thread, out);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(gather48, dim3(grid), dim3(block), 0, 0, (const Rec48 *)buf, uint32_t((size_t(1) << 25) - 1), per_thread, out);   // 32 Mi records = 1.5 GiB
    hipDeviceSynchronize();
    hipLaunchKernelGGL(write32, dim3(grid), dim3(block), 0, 0, (float4 *)buf, bytes / 32);
    hipDeviceSynchronize();
    const double n = double(grid) * block * per_thread;
    std::printf("stream16 bytes=%.0f\ngather8 loads=%.0f sectors64=%.0f bytes_useful=%.0f\ngather48 loads=%.0f bytes_useful=%.0f lines64_touched~=%.0f\nwrite32 bytes=%.0f\n",
                double(bytes), n, n, n * 8, n, n * 48, n * 1.75, double(bytes));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
