// valu_issue_bench.hip -- what is the VALU issue ceiling of one MI355X, in wave64 instructions per second?
//
// bench.py prices a frame's SQ_INSTS_VALU against this ceiling ("valu_issue"); round 3 assumed 4 cycles per wave64 instruction
// (614 G/s) and the C2 record came out at 119 % of it.  This tool measures it: dependence-free streams of one opcode, W waves per
// SIMD (W = 1..8 through the grid size and launch bounds of 64-thread blocks), timed with HIP events and with s_memtime inside
// the kernel (shader cycles), so that both "instructions per second, whole chip" and "cycles per instruction, one SIMD" come out.
// Run under rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE to calibrate what those counters
// count per instruction (tools/valu_calibrate.sh).
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue_bench.hip -o /tmp/valu_issue_bench && /tmp/valu_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

// 64 instructions per loop body, 8 independent accumulators (a wave64 FMA has ~8 cycles of latency: 8 chains cover 2-cycle issue)
enum { OP_FMA = 0, OP_CNDMASK, OP_CMP, OP_PKFMA, OP_RCP, OP_MULLO, OP_DPP, OP_ADD_U32, OP_MIX, OP_SALU, OP_CND_SGPR, OP_CND_ALT, OP_CND_NODEP, OP_VALU_SALU, OP_CMP_VCC, N_OPS };
static const char *op_name[N_OPS] = {"v_fma_f32", "v_cndmask_b32", "v_cmp_lt_f32(sgpr pair)", "v_pk_fma_f32", "v_rcp_f32", "v_mul_lo_u32", "v_mov_b32 dpp row_shr:1", "v_add_u32", "mix fma/cndmask/cmp/add (3:2:1:2)", "s_and_b64 (scalar unit)", "v_cndmask_b32_e64 sgpr-pair mask", "v_cndmask(vcc) / v_add_u32 alternating", "v_cndmask_b32 vcc, dst not a source", "v_add_u32 / s_and_b64 alternating (1:1)", "v_cmp_lt_f32 -> vcc"};

template <int OP>
__global__ __launch_bounds__(64) void stream_kernel(float *out, int iters, unsigned long long *cycles) {
    float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b0 = 1.0001f, b1 = 0.9999f, c0 = 1e-3f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == OP_FMA) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));)
        } else if (OP == OP_CNDMASK) {
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");)
        } else if (OP == OP_CMP) {
            REP8(asm volatile("v_cmp_lt_f32 s[40:41], %0, %8\n v_cmp_lt_f32 s[42:43], %1, %8\n v_cmp_lt_f32 s[44:45], %2, %8\n v_cmp_lt_f32 s[46:47], %3, %8\n"
                              "v_cmp_lt_f32 s[48:49], %4, %8\n v_cmp_lt_f32 s[50:51], %5, %8\n v_cmp_lt_f32 s[52:53], %6, %8\n v_cmp_lt_f32 s[54:55], %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0)
                              : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");)
        } else if (OP == OP_PKFMA) {
            // 4 packed accumulators (register pairs); counts as 8 x 4 = 32 instructions per body pass -> the host halves the count
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1}, r = {c0, c0};
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "v"(r));)
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (OP == OP_RCP) {
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == OP_MULLO) {
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                              "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
        } else if (OP == OP_DPP) {
            REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == OP_ADD_U32) {
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
        } else if (OP == OP_SALU) {
            REP8(asm volatile("s_and_b64 s[40:41], s[56:57], s[58:59]\n s_and_b64 s[42:43], s[56:57], s[58:59]\n s_and_b64 s[44:45], s[56:57], s[58:59]\n s_and_b64 s[46:47], s[56:57], s[58:59]\n"
                              "s_and_b64 s[48:49], s[56:57], s[58:59]\n s_and_b64 s[50:51], s[56:57], s[58:59]\n s_and_b64 s[52:53], s[56:57], s[58:59]\n s_and_b64 s[54:55], s[56:57], s[58:59]\n"
                              : "+v"(a0) : : "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");)
        } else if (OP == OP_CND_SGPR) {
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[56:57]\n v_cndmask_b32_e64 %1, %1, %8, s[58:59]\n v_cndmask_b32_e64 %2, %2, %8, s[56:57]\n v_cndmask_b32_e64 %3, %3, %8, s[58:59]\n"
                              "v_cndmask_b32_e64 %4, %4, %8, s[56:57]\n v_cndmask_b32_e64 %5, %5, %8, s[58:59]\n v_cndmask_b32_e64 %6, %6, %8, s[56:57]\n v_cndmask_b32_e64 %7, %7, %8, s[58:59]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
        } else if (OP == OP_CND_ALT) {
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_add_u32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %8\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_add_u32 %5, %5, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");)
        } else if (OP == OP_CND_NODEP) {
            REP8(asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                              "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0) : "vcc");)
        } else if (OP == OP_VALU_SALU) {
            REP8(asm volatile("v_add_u32 %0, %0, %8\n s_and_b64 s[40:41], s[56:57], s[58:59]\n v_add_u32 %1, %1, %8\n s_and_b64 s[42:43], s[56:57], s[58:59]\n"
                              "v_add_u32 %2, %2, %8\n s_and_b64 s[44:45], s[56:57], s[58:59]\n v_add_u32 %3, %3, %8\n s_and_b64 s[46:47], s[56:57], s[58:59]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");)
        } else if (OP == OP_CMP_VCC) {
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");)
        } else {
            // the traversal kernels' blend: float arithmetic, selects, compares into SGPR pairs, integer adds
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, vcc\n v_fma_f32 %2, %2, %8, %9\n v_cmp_lt_f32 s[40:41], %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_fma_f32 %6, %6, %8, %9\n v_add_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0) : "vcc", "s40", "s41");)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

typedef void (*KernelFn)(float *, int, unsigned long long *);
static KernelFn kernels[N_OPS] = {stream_kernel<OP_FMA>, stream_kernel<OP_CNDMASK>, stream_kernel<OP_CMP>, stream_kernel<OP_PKFMA>, stream_kernel<OP_RCP>,
                                  stream_kernel<OP_MULLO>, stream_kernel<OP_DPP>, stream_kernel<OP_ADD_U32>, stream_kernel<OP_MIX>, stream_kernel<OP_SALU>, stream_kernel<OP_CND_SGPR>,
                                  stream_kernel<OP_CND_ALT>, stream_kernel<OP_CND_NODEP>, stream_kernel<OP_VALU_SALU>, stream_kernel<OP_CMP_VCC>};

int main(int argc, char **argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 20000;
    const int only_w = argc > 2 ? std::atoi(argv[2]) : 0;                     // one occupancy only (the PMC passes)
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    std::printf("# device %s, %d CUs, %d SIMDs, clockRate %d kHz; %d loop passes of 64 instructions per wave\n", prop.gcnArchName, cus, simds, prop.clockRate, iters);
    std::printf("# ceiling if a wave64 VALU instruction takes 2 cycles of its SIMD at the nominal clock: %.1f G wave-inst/s (4 cycles: %.1f)\n",
                simds * (prop.clockRate * 1e3) / 2 / 1e9, simds * (prop.clockRate * 1e3) / 4 / 1e9);
    float *out; unsigned long long *cyc; CHK(hipMalloc(&out, size_t(simds) * 8 * 64 * sizeof(float))); CHK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    std::printf("%-36s %5s %12s %14s %16s %18s\n", "opcode", "waves", "ms", "G wave-inst/s", "cycles/inst/SIMD", "(s_memtime based)");
    const int first_op = argc > 3 ? std::atoi(argv[3]) : 0;
    for (int op = first_op; op < N_OPS; ++op) {
        for (int w = 1; w <= 8; w = w < 4 ? w + 1 : w + 2) {
            if (only_w && w != only_w) continue;
            const int blocks = simds * w;                                    // 64-thread blocks: the dispatcher deals them round-robin over the SIMDs
            const double per_wave = double(iters) * 64;
            hipLaunchKernelGGL(kernels[op], dim3(blocks), dim3(64), 0, 0, out, 200, cyc);           // warm-up
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kernels[op], dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
            CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
            float ms = 0.f; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c = 0; CHK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            const double insts = per_wave * blocks;
            // one wave saw `c` shader cycles for per_wave instructions while w waves shared its SIMD: the SIMD issued w * per_wave in c cycles
            std::printf("%-36s %5d %12.3f %14.1f %16.3f %18.3f\n", op_name[op], w, ms, insts / (ms * 1e-3) / 1e9,
                        (ms * 1e-3) * (prop.clockRate * 1e3) / (per_wave * w), double(c) / (per_wave * w));
        }
    }
    return 0;
}
