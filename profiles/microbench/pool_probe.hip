// pool_probe.hip — does the stream-ordered pool (hipMallocAsync) recycle freed physical memory into allocations of other
// sizes without the fresh-VRAM cost? build: hipcc --offload-arch=gfx950 -O2 pool_probe.hip -o pool_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t GB = 1ull << 30;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipMemPool_t pool;
    CK(hipDeviceGetDefaultMemPool(&pool, 0));
    uint64_t thr = ~0ull;
    CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    void *t[4], *l[6];
    double t0 = now();
    const size_t tsz[4] = {4 * GB + 123456, 4 * GB + 123456, 8 * GB + 777, 8 * GB + 777};
    for (int i = 0; i < 4; i++) CK(hipMallocAsync(&t[i], tsz[i], s));
    CK(hipStreamSynchronize(s));
    printf("pool: 4 temporaries, 24 GB fresh       %8.1f ms\n", now() - t0);
    t0 = now();
    for (int i = 0; i < 4; i++) CK(hipMemsetAsync(t[i], 1, tsz[i], s));
    CK(hipStreamSynchronize(s));
    printf("memset of them                         %8.1f ms\n", now() - t0);
    t0 = now();
    for (int i = 0; i < 4; i++) CK(hipFreeAsync(t[i], s));
    CK(hipStreamSynchronize(s));
    printf("hipFreeAsync of them                   %8.1f ms\n", now() - t0);
    t0 = now();
    const size_t lsz[6] = {12 * GB + 600000000, 6 * GB + 300000000, GB + 570000000, 12 * GB + 600000000, 6 * GB + 300000000, GB + 570000000};
    for (int i = 0; i < 3; i++) CK(hipMallocAsync(&l[i], lsz[i], s));
    CK(hipStreamSynchronize(s));
    printf("pool: 20.5 GB of other sizes (reuse)   %8.1f ms\n", now() - t0);
    t0 = now();
    for (int i = 0; i < 3; i++) CK(hipMemsetAsync(l[i], 1, lsz[i], s));
    CK(hipStreamSynchronize(s));
    printf("memset of them                         %8.1f ms\n", now() - t0);
    t0 = now();
    for (int i = 3; i < 6; i++) CK(hipMallocAsync(&l[i], lsz[i], s));
    CK(hipStreamSynchronize(s));
    printf("pool: 20.5 GB more (3.5 reuse + fresh) %8.1f ms\n", now() - t0);
    t0 = now();
    for (int i = 3; i < 6; i++) CK(hipMemsetAsync(l[i], 1, lsz[i], s));
    CK(hipStreamSynchronize(s));
    printf("memset of them                         %8.1f ms\n", now() - t0);
    void *p;
    t0 = now();
    CK(hipMalloc(&p, 8 * GB));
    CK(hipMemset(p, 1, 8 * GB));
    CK(hipDeviceSynchronize());
    printf("plain hipMalloc 8 GB + memset          %8.1f ms\n", now() - t0);
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GB of %.1f\n", fr / 1e9, tot / 1e9);
    return 0;
}
