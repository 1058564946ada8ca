// VALU issue-rate microbenchmark for the ops of the DP inner loop (gfx950).  Build + run:
//   hipcc --offload-arch=gfx950 -O3 -o valu_ubench tools/valu_ubench.hip && ./valu_ubench
// Each kernel runs ITER iterations of 32 back-to-back independent copies of one instruction (8 accumulators
// round-robin), one wave per block, blocks = 256 CUs x 4 SIMDs x W waves.  Reports cycles per wave-instruction
// per SIMD, assuming the measured clock of a v_add_u32 calibration at 2 cycles... we just print ns and derive.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFK(NAME, ASM)                                                                     \
    __global__ __launch_bounds__(64) void NAME(int iters, int *out, int c0, int c1) {       \
        int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;                                   \
        for (int i = 0; i < iters; i++) {                                                   \
            asm volatile(ASM ASM ASM ASM                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b0), "v"(b1), "s"(c0), "s"(c1));                             \
        }                                                                                   \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;                  \
    }

// %0..%7 accumulators, %8 %9 vgpr inputs, %10 %11 sgpr inputs
DEFK(k_add, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n")
DEFK(k_add_s, "v_add_u32 %0, %10, %0\n v_add_u32 %1, %10, %1\n v_add_u32 %2, %10, %2\n v_add_u32 %3, %10, %3\n v_add_u32 %4, %11, %4\n v_add_u32 %5, %11, %5\n v_add_u32 %6, %11, %6\n v_add_u32 %7, %11, %7\n")
DEFK(k_or_imm, "v_or_b32 %0, 3, %0\n v_or_b32 %1, 3, %1\n v_or_b32 %2, 3, %2\n v_or_b32 %3, 3, %3\n v_or_b32 %4, 3, %4\n v_or_b32 %5, 3, %5\n v_or_b32 %6, 3, %6\n v_or_b32 %7, 3, %7\n")
DEFK(k_max, "v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %9\n v_max_i32 %5, %5, %9\n v_max_i32 %6, %6, %9\n v_max_i32 %7, %7, %9\n")
DEFK(k_max3, "v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %1, %1, %8, %9\n v_max3_i32 %2, %2, %8, %9\n v_max3_i32 %3, %3, %8, %9\n v_max3_i32 %4, %4, %9, %8\n v_max3_i32 %5, %5, %9, %8\n v_max3_i32 %6, %6, %9, %8\n v_max3_i32 %7, %7, %9, %8\n")
DEFK(k_max3_2v, "v_max3_i32 %0, %0, %8, %8\n v_max3_i32 %1, %1, %8, %8\n v_max3_i32 %2, %2, %8, %8\n v_max3_i32 %3, %3, %8, %8\n v_max3_i32 %4, %4, %9, %9\n v_max3_i32 %5, %5, %9, %9\n v_max3_i32 %6, %6, %9, %9\n v_max3_i32 %7, %7, %9, %9\n")
DEFK(k_andor, "v_and_or_b32 %0, %0, -4, 2\n v_and_or_b32 %1, %1, -4, 2\n v_and_or_b32 %2, %2, -4, 2\n v_and_or_b32 %3, %3, -4, 2\n v_and_or_b32 %4, %4, -4, 1\n v_and_or_b32 %5, %5, -4, 1\n v_and_or_b32 %6, %6, -4, 1\n v_and_or_b32 %7, %7, -4, 1\n")
DEFK(k_alignbit, "v_alignbit_b32 %0, %8, %0, 2\n v_alignbit_b32 %1, %8, %1, 2\n v_alignbit_b32 %2, %8, %2, 2\n v_alignbit_b32 %3, %8, %3, 2\n v_alignbit_b32 %4, %9, %4, 2\n v_alignbit_b32 %5, %9, %5, 2\n v_alignbit_b32 %6, %9, %6, 2\n v_alignbit_b32 %7, %9, %7, 2\n")
DEFK(k_add_sdwa, "v_add_u32_sdwa %0, sext(%8), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32_sdwa %1, sext(%8), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, sext(%8), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32_sdwa %3, sext(%8), %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %4, sext(%9), %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32_sdwa %5, sext(%9), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %6, sext(%9), %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32_sdwa %7, sext(%9), %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
DEFK(k_dpp, "v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n")
DEFK(k_add_dpp, "v_add_u32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %9, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %9, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %9, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %9, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
DEFK(k_fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %9, %8\n v_fma_f32 %5, %5, %9, %8\n v_fma_f32 %6, %6, %9, %8\n v_fma_f32 %7, %7, %9, %8\n")
DEFK(k_pk_add_i16, "v_pk_add_i16 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_pk_add_i16 %2, %2, %8\n v_pk_add_i16 %3, %3, %8\n v_pk_add_i16 %4, %4, %9\n v_pk_add_i16 %5, %5, %9\n v_pk_add_i16 %6, %6, %9\n v_pk_add_i16 %7, %7, %9\n")
DEFK(k_pk_max_i16, "v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %9\n v_pk_max_i16 %5, %5, %9\n v_pk_max_i16 %6, %6, %9\n v_pk_max_i16 %7, %7, %9\n")
DEFK(k_bfi, "v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n v_bfi_b32 %4, %9, %4, %8\n v_bfi_b32 %5, %9, %5, %8\n v_bfi_b32 %6, %9, %6, %8\n v_bfi_b32 %7, %9, %7, %8\n")
DEFK(k_add3, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %9, %8\n v_add3_u32 %5, %5, %9, %8\n v_add3_u32 %6, %6, %9, %8\n v_add3_u32 %7, %7, %9, %8\n")
DEFK(k_dep_add, "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %9\n v_add_u32 %0, %0, %9\n v_add_u32 %0, %0, %9\n v_add_u32 %0, %0, %9\n")
DEFK(k_dep_max3, "v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %0, %0, %9, %8\n v_max3_i32 %0, %0, %9, %8\n v_max3_i32 %0, %0, %9, %8\n v_max3_i32 %0, %0, %9, %8\n")

typedef void (*kfn)(int, int *, int, int);
struct K { const char *name; kfn f; };

int main() {
    std::vector<K> ks = {{"v_add_u32 (vv)", k_add}, {"v_add_u32 (sv)", k_add_s}, {"v_or_b32 imm", k_or_imm}, {"v_max_i32", k_max}, {"v_max3_i32 (3 vgpr)", k_max3},
                         {"v_max3_i32 (2 distinct vgpr)", k_max3_2v}, {"v_and_or_b32 imm", k_andor}, {"v_alignbit_b32", k_alignbit},
                         {"v_add_u32_sdwa sext", k_add_sdwa}, {"v_mov_b32_dpp", k_dpp}, {"v_add_u32_dpp", k_add_dpp}, {"v_fma_f32", k_fma},
                         {"v_pk_add_i16", k_pk_add_i16}, {"v_pk_max_i16", k_pk_max_i16}, {"v_bfi_b32", k_bfi}, {"v_add3_u32", k_add3},
                         {"dependent v_add_u32 chain", k_dep_add}, {"dependent v_max3_i32 chain", k_dep_max3}};
    int *out; hipMalloc(&out, 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int W : {1, 2, 4, 8}) {
        printf("--- %d wave(s) per SIMD ---\n", W);
        for (auto &k : ks) {
            const int blocks = cus * 4 * W;
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(64), 0, 0, 100, out, 1, 2);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(64), 0, 0, iters, out, 1, 2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_simd = (double)iters * 32 * W;
            const double ns_per_inst = ms * 1e6 / insts_per_simd;
            printf("%-32s %8.3f ms  %.3f ns/inst/SIMD  = %.2f cyc @2.4GHz, %.2f cyc @2.1GHz\n", k.name, ms, ns_per_inst, ns_per_inst * 2.4, ns_per_inst * 2.1);
        }
    }
    return 0;
}
