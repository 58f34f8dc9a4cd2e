// alloc_probe2.hip — which call pays for recycled VRAM: the hipFree, the next hipMalloc, or the first touch? And does waiting help?
// Mimics a handle's life: 8 x 8 GB allocated, written, freed; then the same again (a) at once, (b) after a pause, (c) as one 64 GB block.
// build: hipcc --offload-arch=gfx950 -O2 alloc_probe2.hip -o alloc_probe2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t GB = 1ull << 30;
static void round_of(const char *tag, int n, size_t each, void **p) {
    double tot = 0, worst = 0;
    for (int i = 0; i < n; i++) {
        double t = now();
        if (hipMalloc(&p[i], each) != hipSuccess) { printf("hipMalloc failed\n"); return; }
        double d = now() - t;
        tot += d;
        if (d > worst) worst = d;
        if (d > 5.0) printf("   %s: hipMalloc #%d of %zu GB took %.1f ms\n", tag, i, each / GB, d);
    }
    printf("%-44s mallocs %8.1f ms (worst %.1f)\n", tag, tot, worst);
    double t = now();
    for (int i = 0; i < n; i++) hipMemsetAsync(p[i], 1, each, 0);
    hipDeviceSynchronize();
    printf("%-44s memset  %8.1f ms\n", tag, now() - t);
}
static void free_all(const char *tag, int n, void **p) {
    double t = now();
    for (int i = 0; i < n; i++) hipFree(p[i]);
    printf("%-44s frees   %8.1f ms\n", tag, now() - t);
}
int main() {
    void *p[16];
    hipFree(0);
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    round_of("fresh: 8 x 8 GB", 8, 8 * GB, p);
    free_all("fresh: 8 x 8 GB", 8, p);
    round_of("right after the frees: 8 x 8 GB", 8, 8 * GB, p);
    free_all("second", 8, p);
    std::this_thread::sleep_for(std::chrono::seconds(5));
    round_of("5 s after the frees: 8 x 8 GB", 8, 8 * GB, p);
    free_all("third", 8, p);
    round_of("right after: ONE 64 GB block", 1, 64 * GB, p);
    free_all("fourth", 1, p);
    round_of("right after: 16 x 4 GB", 16, 4 * GB, p);
    // half freed, half kept, then new allocations of another size (the transposed build's pattern: temporaries freed, layouts allocated)
    free_all("free the first 8 of the 16", 8, p);
    void *q[4];
    round_of("then 4 x 6 GB", 4, 6 * GB, q);
    free_all("rest", 8, p + 8);
    free_all("rest", 4, q);
    hipMemGetInfo(&fr, &tot);
    printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    return 0;
}
