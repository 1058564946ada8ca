// How many one-wave workgroups with a given LDS size does a CU of the MI355X hold at once?  (The compact LDS layouts of round 3 rest on
// the answer: LDS is handed out in granules, and the kernels' occupancy follows the granule count, not the byte count.)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/lds_occupancy.bin tools/lds_occupancy.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void hold_kernel(int *resident, int *peak, long long ticks) {
    extern __shared__ int lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1;
        const int r = atomicAdd(resident, 1) + 1;
        atomicMax(peak, r);
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
        atomicSub(resident, 1);
    }
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    int *d;
    hipMalloc(&d, 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = {1024, 4096, 5120, 5121, 6400, 6401, 6528, 7680, 7681, 8064, 8192, 8960, 8961, 10240, 10241, 12800, 12801, 13312, 13440, 13824, 14080, 14081, 16000, 16384, 16640, 16641, 32768, 65536};
    printf("CUs %d, LDS per CU %zu\n", cus, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    for (int sz : sizes) {
        hipMemset(d, 0, 8);
        hipLaunchKernelGGL(hold_kernel, dim3(cus * 40), dim3(64), sz, 0, d, d + 1, 20000LL /* 200 us at 100 MHz */);
        hipDeviceSynchronize();
        int h[2];
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("lds %6d B: peak resident %5d = %.2f per CU   (160 KB / size = %.2f; by 1280-B granules %d, by 512-B granules %d)\n", sz, h[1], (double)h[1] / cus, 163840.0 / sz,
               163840 / ((sz + 1279) / 1280 * 1280), 163840 / ((sz + 511) / 512 * 512));
    }
    return 0;
}
