// LDS read-rate microbenchmark (gfx950): how expensive are 20 ds_read_i16 vs 10 ds_read2_b32 vs 5 ds_read_b128 per "step"
// when interleaved with ~200 cycles of VALU, at 1..3 waves per SIMD?  Build: hipcc --offload-arch=gfx950 -O3 -o lds_ubench.bin tools/lds_ubench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE, int NVALU>
__global__ __launch_bounds__(64) void k(int iters, int *out, int stride) {
    __shared__ int lds[3072]; // 12 KB per wave
    const int lane = threadIdx.x;
    for (int i = lane; i < 3072; i += 64) lds[i] = i;
    __syncthreads();
    int acc[8] = {lane, 1, 2, 3, 4, 5, 6, 7};
    int off = (lane * (MODE == 2 ? 4 : 1) * stride) & 1023; // conflict-free: dword stride 1 (b32/i16) or 4 (b128)
    for (int it = 0; it < iters; it++) {
        const char *base = reinterpret_cast<const char *>(lds) + off * 4;
        int v[20];
        const unsigned a = (unsigned)(off * 4) + (unsigned)(size_t)lds; // LDS byte address (low 32 bits of the shared pointer)
        if (MODE == 0) {
#define RD16(r) asm volatile("ds_read_i16 %0, %1 offset:" #r : "=v"(v[(r) / 2]) : "v"(a));
            RD16(0) RD16(2) RD16(4) RD16(6) RD16(8) RD16(10) RD16(12) RD16(14) RD16(16) RD16(18) RD16(20) RD16(22) RD16(24) RD16(26) RD16(28) RD16(30) RD16(32) RD16(34) RD16(36) RD16(38)
        } else if (MODE == 1) {
#define RD2(r, o0, o1) asm volatile("ds_read2_b32 %0, %1 offset0:" #o0 " offset1:" #o1 : "=v"(*reinterpret_cast<long long *>(&v[r])) : "v"(a));
            RD2(0, 0, 1) RD2(2, 2, 3) RD2(4, 4, 5) RD2(6, 6, 7) RD2(8, 8, 9) RD2(10, 10, 11) RD2(12, 12, 13) RD2(14, 14, 15) RD2(16, 16, 17) RD2(18, 18, 19)
        } else if (MODE == 2) {
#define RD4(r, o) asm volatile("ds_read_b128 %0, %1 offset:" #o : "=v"(*reinterpret_cast<int4 *>(&v[r])) : "v"(a));
            RD4(0, 0) RD4(4, 16) RD4(8, 32) RD4(12, 48) RD4(16, 64)
        } else {
#pragma unroll
            for (int r = 0; r < 20; r++) v[r] = r;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < NVALU; q++) acc[q & 7] += v[q % 20] + acc[(q + 1) & 7];
    }
    int s = 0;
    for (int q = 0; q < 8; q++) s += acc[q];
    if (s == 0x1234567) out[0] = s;
}
template <int MODE, int NVALU> void run(const char *name, int W, int *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 4 * W;
    hipLaunchKernelGGL((k<MODE, NVALU>), dim3(blocks), dim3(64), 0, 0, 10, out, 1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NVALU>), dim3(blocks), dim3(64), 0, 0, iters, out, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s W=%d NVALU=%3d  %.1f ns per step per wave-slot  (%.0f cyc @2.1GHz per SIMD-step)\n", name, W, NVALU, ms * 1e6 / iters, ms * 1e6 / iters / W * 2.1);
}
int main() {
    int *out; hipMalloc(&out, 4);
    for (int W : {1, 2, 3}) {
        run<3, 100>("no LDS", W, out);
        run<0, 100>("20 x ds_read_i16", W, out);
        run<1, 100>("20 x b32 (ds_read2_b32 x10)", W, out);
        run<2, 100>("5 x ds_read_b128", W, out);
        run<3, 200>("no LDS", W, out);
        run<0, 200>("20 x ds_read_i16", W, out);
        run<1, 200>("20 x b32 (ds_read2_b32 x10)", W, out);
        run<2, 200>("5 x ds_read_b128", W, out);
    }
    return 0;
}
