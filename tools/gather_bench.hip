// tools/gather_bench.hip -- what the MI355X memory system sustains for the traversal's access pattern: every lane follows a
// DEPENDENT chain of random gathers (the next index is a hash of the loaded word) in a table of `mb` megabytes.
//   mode 0: 8-byte gathers (one kd node)         mode 1: 16-byte gathers (a sibling pair)
//   mode 2: 64-byte gathers as 4 x dwordx4 (a whole treelet line)    mode 4: 128-byte aligned gathers (4 x dwordx4 spread over both halves)    mode 3: 8-byte gathers, two independent chains per lane
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o gpurun_out/gather_bench ; run: gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ unsigned h32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(256) void chase(const uint4 *__restrict__ tab, unsigned n_lines, int steps, unsigned *out) {
    unsigned i = h32(blockIdx.x * 256 + threadIdx.x + 1u);
    unsigned j = h32(i);
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        const unsigned line = i % n_lines;                       // 64-byte line
        if (MODE == 0) { const uint2 v = *((const uint2 *)(tab + size_t(line) * 4) + (i >> 29)); acc += v.y; i = h32(v.x + i); }
        else if (MODE == 1) { const uint4 v = tab[size_t(line) * 4 + (i >> 30)]; acc += v.y + v.w; i = h32(v.x + i); }
        else if (MODE == 4) { const unsigned l128 = line & ~1u; const uint4 a = tab[size_t(l128) * 4], b = tab[size_t(l128) * 4 + 2], c = tab[size_t(l128) * 4 + 5], d = tab[size_t(l128) * 4 + 7];
                              acc += a.y + b.y + c.y + d.w; i = h32(a.x + b.x + c.x + d.x + i); }
        else if (MODE == 2) { const uint4 a = tab[size_t(line) * 4], b = tab[size_t(line) * 4 + 1], c = tab[size_t(line) * 4 + 2], d = tab[size_t(line) * 4 + 3];
                              acc += a.y + b.y + c.y + d.w; i = h32(a.x + b.x + c.x + d.x + i); }
        else { const unsigned l2 = j % n_lines; const uint2 v = *((const uint2 *)(tab + size_t(line) * 4) + (i >> 29)); const uint2 w = *((const uint2 *)(tab + size_t(l2) * 4) + (j >> 29));
               acc += v.y + w.y; i = h32(v.x + i); j = h32(w.x + j); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + i + j;
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned *out; hipMalloc(&out, size_t(cus) * 8 * 256 * 4);
    for (int kb : {16, 256, 2048, 16384, 307200}) {
        const int mb = kb >> 10; const size_t bytes = size_t(kb) << 10; const unsigned n_lines = unsigned(bytes / 64);
        uint4 *tab; hipMalloc(&tab, bytes);
        std::vector<unsigned> h(bytes / 4); unsigned s = 12345; for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s; }
        hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice);
        for (int mode : {0, 2, 4})
            for (int bpc : {1, 2, 4}) {
                const int steps = 2000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                auto launch = [&](int st) {
                    if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(cus * bpc), dim3(256), 0, 0, tab, n_lines, st, out);
                    if (mode == 1) hipLaunchKernelGGL(chase<1>, dim3(cus * bpc), dim3(256), 0, 0, tab, n_lines, st, out);
                    if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(cus * bpc), dim3(256), 0, 0, tab, n_lines, st, out);
                    if (mode == 4) hipLaunchKernelGGL(chase<4>, dim3(cus * bpc), dim3(256), 0, 0, tab, n_lines, st, out);
                    if (mode == 3) hipLaunchKernelGGL(chase<3>, dim3(cus * bpc), dim3(256), 0, 0, tab, n_lines, st, out); };
                launch(50); hipDeviceSynchronize();
                hipEventRecord(e0); launch(steps); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double lanes = double(cus) * bpc * 256, gathers = lanes * steps * (mode == 3 ? 2 : 1);
                printf("table %6d KB mode %d blocks/CU %d: %.2f ms  %.1f G line-gathers/s  (%.2f TB/s of 64-B lines)  %.0f ns per dependent step\n", kb, mode, bpc, ms,
                       gathers / ms / 1e6, gathers * 64 / ms / 1e9, ms * 1e6 / steps);
            }
        hipFree(tab);
    }
    return 0;
}
