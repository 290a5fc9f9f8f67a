// GPU box: do two kernels on two CU-masked streams run side by side?  Which CUs does a mask select on an 8-XCD part?
// build: hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <chrono>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(unsigned *ids, unsigned long long cycles, float *sink) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, offset 0, size 32
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID[3:0]
    if (threadIdx.x == 0) ids[blockIdx.x] = (xcc << 16) | (hw & 0xffffu);
    const unsigned long long t0 = __builtin_readcyclecounter();
    float a = threadIdx.x;
    while (__builtin_readcyclecounter() - t0 < cycles) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) *sink = a;
}

static void summarize(const char *tag, const std::vector<unsigned> &ids) {
    std::set<unsigned> cus; std::set<unsigned> xccs;
    for (unsigned v : ids) { const unsigned xcc = v >> 16, hw = v & 0xffff; const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc); }
    printf("%s: %zu blocks on %zu distinct (xcc,se,sh,cu), %zu xccs\n", tag, ids.size(), cus.size(), xccs.size());
}

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("CUs %d clock %d kHz\n", ncu, prop.clockRate);
    unsigned *ids1, *ids2; float *sink;
    const int G = 4096;
    CHK(hipMalloc(&ids1, G * 4)); CHK(hipMalloc(&ids2, G * 4)); CHK(hipMalloc(&sink, 4));
    for (int split : {128, 192, 224}) {
        for (int layout = 0; layout < 2; ++layout) {
            // layout 0: trace = bits [0, split), shade = [split, ncu).  layout 1: bits interleaved (bit i belongs to shade when i % 8 >= 8 * split / ncu)
            std::vector<uint32_t> m1(ncu / 32, 0), m2(ncu / 32, 0);
            for (int i = 0; i < ncu; ++i) {
                const bool first = layout == 0 ? i < split : (i % 8) < (8 * split / ncu);
                (first ? m1 : m2)[i / 32] |= 1u << (i % 32);
            }
            hipStream_t s1, s2;
            CHK(hipExtStreamCreateWithCUMask(&s1, m1.size(), m1.data()));
            CHK(hipExtStreamCreateWithCUMask(&s2, m2.size(), m2.data()));
            const unsigned long long cyc = 100000000ull / 10;     // ~0.1 s at 100 MHz counter
            // alone
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, s1, ids1, cyc / 64, sink); CHK(hipStreamSynchronize(s1));
            auto t1 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, s2, ids2, cyc / 64, sink); CHK(hipStreamSynchronize(s2));
            auto t2 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, s1, ids1, cyc / 64, sink);
            hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, s2, ids2, cyc / 64, sink);
            CHK(hipStreamSynchronize(s1)); CHK(hipStreamSynchronize(s2));
            auto t3 = std::chrono::steady_clock::now();
            std::vector<unsigned> h1(G), h2(G);
            CHK(hipMemcpy(h1.data(), ids1, G * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h2.data(), ids2, G * 4, hipMemcpyDeviceToHost));
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            printf("split %d layout %d: s1 alone %.2f ms, s2 alone %.2f ms, both %.2f ms\n", split, layout, ms(t0, t1), ms(t1, t2), ms(t2, t3));
            summarize("  s1", h1); summarize("  s2", h2);
            std::set<unsigned> a, b; for (unsigned v : h1) a.insert(v >> 4 & 0xfffff0 | (v >> 8 & 15)); for (unsigned v : h2) b.insert(v >> 4 & 0xfffff0 | (v >> 8 & 15));
            size_t common = 0; for (unsigned v : a) common += b.count(v);
            printf("  CUs in common: %zu\n", common);
            CHK(hipStreamDestroy(s1)); CHK(hipStreamDestroy(s2));
        }
    }
    return 0;
}
