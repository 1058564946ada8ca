// How many T-thread workgroups with a given LDS size and register count does a CU hold at once?  (const_long_wg.hip.h: a workgroup of
// 5 waves fragments the SIMDs' wave slots.)  Build: hipcc --offload-arch=gfx950 -O2 -o tools/wg_occupancy.bin tools/wg_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_num_vgpr(96))) void hold_kernel(int *resident, int *peak, long long ticks) {
    extern __shared__ int lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1;
        const int r = atomicAdd(resident, 1) + 1;
        atomicMax(peak, r);
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
        atomicSub(resident, 1);
    }
    __syncthreads();
}
template <int T>
void run(int cus, int *d, int sz) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hold_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL(hold_kernel<T>, dim3(cus * 24), dim3(T), sz, 0, d, d + 1, 20000LL);
    hipDeviceSynchronize();
    int h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("threads %4d lds %6d B (96 VGPRs): peak resident workgroups %5d = %.2f per CU = %.2f waves per SIMD\n", T, sz, h[1], (double)h[1] / cus, (double)h[1] / cus * (T / 64) / 4.0);
}
int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    int *d;
    hipMalloc(&d, 8);
    for (int sz : {6528, 31928, 32000, 32952, 40896, 40960, 62664}) { run<64>(cus, d, sz); run<256>(cus, d, sz); run<320>(cus, d, sz); run<512>(cus, d, sz); run<640>(cus, d, sz); }
    return 0;
}
