// alloc_probe.hip — cost of hipMalloc / hipFree of multi-GB buffers on MI355X (first-call analysis: a 2.5 s "tile layout
// build" was an allocation waiting for freshly freed VRAM). build: hipcc --offload-arch=gfx950 -O2 alloc_probe.hip -o alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t GB = 1ull << 30;
    void *a = nullptr, *b = nullptr, *c = nullptr;
    hipFree(0);
    double t = now();
    hipMalloc(&a, 12 * GB);
    printf("fresh hipMalloc 12 GB            %8.1f ms\n", now() - t);
    t = now();
    hipMemset(a, 1, 12 * GB);
    hipDeviceSynchronize();
    printf("memset 12 GB                     %8.1f ms\n", now() - t);
    t = now();
    hipFree(a);
    printf("hipFree 12 GB                    %8.1f ms\n", now() - t);
    t = now();
    hipMalloc(&a, 12 * GB);
    printf("hipMalloc 12 GB right after free %8.1f ms\n", now() - t);
    t = now();
    hipMalloc(&b, 20 * GB);
    printf("fresh hipMalloc 20 GB            %8.1f ms\n", now() - t);
    hipMemset(b, 1, 20 * GB);
    hipDeviceSynchronize();
    t = now();
    hipFree(a);
    hipFree(b);
    printf("hipFree 12 + 20 GB               %8.1f ms\n", now() - t);
    t = now();
    hipMalloc(&c, 27 * GB);
    printf("hipMalloc 27 GB after the frees  %8.1f ms\n", now() - t);
    t = now();
    hipMemset(c, 1, 27 * GB);
    hipDeviceSynchronize();
    printf("memset 27 GB                     %8.1f ms\n", now() - t);
    hipFree(c);
    for (int i = 0; i < 3; i++) {
        t = now();
        hipMalloc(&c, 1 * GB);
        double t1 = now() - t;
        t = now();
        hipFree(c);
        printf("hipMalloc / hipFree 1 GB         %8.1f / %.1f ms\n", t1, now() - t);
    }
    return 0;
}
