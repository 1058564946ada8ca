// alloc_probe.hip -- what a device allocation costs on this system, by API (round 6): hipMalloc, hipMallocAsync, hipExtMallocWithFlags,
// virtual memory management (reserve + create + map), two threads at once, and whether the cost is paid at the call or at the first touch.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/alloc_probe.bin tools/alloc_probe.hip ; run: tools/alloc_probe.bin [GB per trial]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static int g_fail = 0;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_fail++; printf("  %s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(unsigned *p, size_t n_pages, size_t stride_dw) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n_pages) p[i * stride_dw] = 1u; }
static double touch_ms(void *p, size_t bytes) {
    if (!p || g_fail) { g_fail = 0; return -1.0; }
    const size_t pages = bytes / 4096;
    const double t0 = now();
    hipLaunchKernelGGL(touch, dim3((unsigned)((pages + 255) / 256)), dim3(256), 0, 0, (unsigned *)p, pages, (size_t)1024);
    CK(hipDeviceSynchronize());
    return (now() - t0) * 1e3;
}
int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 16.0;
    const size_t bytes = (size_t)(gb * 1e9) & ~((size_t)(2 << 20) - 1);
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot)); printf("free %.1f GB of %.1f\n", fr / 1e9, tot / 1e9);
    for (int rep = 0; rep < 2; rep++) {
        void *p = nullptr; double t0 = now(); CK(hipMalloc(&p, bytes)); double t1 = now();
        printf("hipMalloc %.0f GB: %.1f ms (%.2f ms/GB), first touch of every page %.1f ms", gb, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / gb, touch_ms(p, bytes));
        printf(", second touch %.1f ms", touch_ms(p, bytes));
        t0 = now(); CK(hipFree(p)); printf(", hipFree %.1f ms\n", (now() - t0) * 1e3);
    }
    for (double g : {0.25, 1.0, 4.0}) {
        const size_t b = (size_t)(g * 1e9); const int k = (int)(gb / g);
        std::vector<void *> ps((size_t)k, nullptr); double t0 = now(); for (auto &q : ps) CK(hipMalloc(&q, b)); double t1 = now();
        printf("%d x hipMalloc %.2f GB: %.1f ms (%.2f ms/GB)\n", k, g, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / (k * g));
        for (auto q : ps) CK(hipFree(q));
    }
    { // stream-ordered pool
        hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        for (int rep = 0; rep < 2; rep++) {
            void *p = nullptr; double t0 = now(); CK(hipMallocAsync(&p, bytes, 0)); CK(hipStreamSynchronize(0)); double t1 = now();
            printf("hipMallocAsync %.0f GB (%s): %.1f ms (%.2f ms/GB), touch %.1f ms", gb, rep ? "pool holds it" : "fresh", (t1 - t0) * 1e3, (t1 - t0) * 1e3 / gb, touch_ms(p, bytes));
            t0 = now(); CK(hipFreeAsync(p, 0)); CK(hipStreamSynchronize(0)); printf(", hipFreeAsync %.1f ms\n", (now() - t0) * 1e3);
        }
        CK(hipMemPoolTrimTo(pool, 0));
    }
    for (unsigned flag : {(unsigned)hipDeviceMallocUncached, (unsigned)hipDeviceMallocFinegrained}) {
        void *p = nullptr; double t0 = now(); CK(hipExtMallocWithFlags(&p, bytes, flag)); double t1 = now();
        printf("hipExtMallocWithFlags(%s) %.0f GB: %.1f ms (%.2f ms/GB), touch %.1f ms\n", flag == hipDeviceMallocUncached ? "uncached" : "fine-grained", gb, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / gb, p ? touch_ms(p, bytes) : 0.0);
        if (p) CK(hipFree(p));
    }
    { // virtual memory management: reserve the range once, create + map chunk by chunk
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended)); printf("VMM granularity %zu\n", gran);
        for (double cg : {2.0, 0.25}) {
            const size_t chunk = ((size_t)(cg * 1e9) + gran - 1) / gran * gran; const int k = (int)(bytes / chunk);
            void *va = nullptr; double t0 = now(); CK(hipMemAddressReserve(&va, chunk * k, 0, nullptr, 0)); double t_res = now() - t0;
            std::vector<hipMemGenericAllocationHandle_t> hs((size_t)k);
            double t_create = 0, t_map = 0, t_acc = 0;
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            for (int i = 0; i < k; i++) {
                t0 = now(); CK(hipMemCreate(&hs[(size_t)i], chunk, &prop, 0)); t_create += now() - t0;
                t0 = now(); CK(hipMemMap((char *)va + (size_t)i * chunk, chunk, 0, hs[(size_t)i], 0)); t_map += now() - t0;
                t0 = now(); CK(hipMemSetAccess((char *)va + (size_t)i * chunk, chunk, &acc, 1)); t_acc += now() - t0;
            }
            const double g_tot = chunk * (double)k / 1e9;
            printf("VMM %d chunks of %.2f GB: reserve %.2f ms, create %.1f ms (%.2f ms/GB), map %.1f ms, set access %.1f ms (%.2f ms/GB), touch %.1f ms\n", k, chunk / 1e9, t_res * 1e3, t_create * 1e3, t_create * 1e3 / g_tot,
                   t_map * 1e3, t_acc * 1e3, t_acc * 1e3 / g_tot, touch_ms(va, chunk * k));
            t0 = now();
            for (int i = 0; i < k; i++) { CK(hipMemUnmap((char *)va + (size_t)i * chunk, chunk)); CK(hipMemRelease(hs[(size_t)i])); }
            CK(hipMemAddressFree(va, chunk * k)); printf("  unmap + release %.1f ms\n", (now() - t0) * 1e3);
        }
    }
    { // two host threads allocating at once
        void *p[2] = {nullptr, nullptr}; const double t0 = now();
        std::thread a([&] { CK(hipSetDevice(0)); CK(hipMalloc(&p[0], bytes / 2)); }), b([&] { CK(hipSetDevice(0)); CK(hipMalloc(&p[1], bytes / 2)); });
        a.join(); b.join();
        printf("two threads x hipMalloc %.0f GB: %.1f ms (%.2f ms/GB)\n", gb / 2, (now() - t0) * 1e3, (now() - t0) * 1e3 / gb);
        CK(hipFree(p[0])); CK(hipFree(p[1]));
    }
    { // the same API again and again, with pauses: is a slow allocation one that got memory freed a moment ago (cleared in the background)?
        for (double g2 : {32.0, 128.0}) {
            const size_t b2 = (size_t)(g2 * 1e9);
            for (int rep = 0; rep < 3; rep++) {
                void *p = nullptr; double t0 = now(); CK(hipMalloc(&p, b2)); double t1 = now(); const double tt = touch_ms(p, b2); double t2 = now(); CK(hipFree(p));
                printf("again: hipMalloc %.0f GB %.1f ms, touch %.1f ms, hipFree %.1f ms\n", g2, (t1 - t0) * 1e3, tt, (now() - t2) * 1e3);
            }
            std::this_thread::sleep_for(std::chrono::seconds(4));
            { void *p = nullptr; double t0 = now(); CK(hipMalloc(&p, b2)); printf("after 4 s idle: hipMalloc %.0f GB %.1f ms\n", g2, (now() - t0) * 1e3); CK(hipFree(p)); }
            for (int rep = 0; rep < 3; rep++) {
                void *p = nullptr; double t0 = now(); CK(hipMallocAsync(&p, b2, 0)); CK(hipStreamSynchronize(0)); double t1 = now(); const double tt = touch_ms(p, b2);
                double t2 = now(); CK(hipFreeAsync(p, 0)); CK(hipStreamSynchronize(0));
                printf("again: hipMallocAsync %.0f GB %.1f ms, touch %.1f ms, hipFreeAsync %.1f ms\n", g2, (t1 - t0) * 1e3, tt, (now() - t2) * 1e3);
            }
            { hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0)); CK(hipMemPoolTrimTo(pool, 0)); }
        }
    }
    return 0;
}
