// microbenchmark: LDS f64 atomic add / read / RMW throughput with row-contiguous wave accesses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int ROWS = 128;       // rows of 1024 B
constexpr int ITERS = 8192;
template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, unsigned seed) {
    __shared__ double lds[ROWS * 128];
    const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (unsigned i = tid; i < ROWS * 128; i += 1024) lds[i] = 0.0;
    __syncthreads();
    unsigned s = seed + w * 7919u + blockIdx.x * 104729u;
    double acc0 = 0, acc1 = 0;
    for (int it = 0; it < ITERS; it++) {
        s = s * 1664525u + 1013904223u;
        const unsigned row = (s >> 16) % ROWS;   // wave-uniform
        if (MODE == 0) {        // two ds_add_f64 (64 lanes x 8 B each = 1 KB row)
            __hip_atomic_fetch_add(&lds[row * 128 + lane], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lds[row * 128 + 64 + lane], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 1) { // ds_read_b128 only
            d2 x = *reinterpret_cast<d2 *>(&lds[row * 128 + lane * 2]);
            acc0 += x.x; acc1 += x.y;
        } else if (MODE == 2) { // read b128 + write b128 (non-atomic RMW)
            d2 x = *reinterpret_cast<d2 *>(&lds[row * 128 + lane * 2]);
            x.x += 1.0; x.y += 1.0;
            *reinterpret_cast<d2 *>(&lds[row * 128 + lane * 2]) = x;
        } else {                // read X row (b128) + 2 atomic adds to another row: the proposed inner step
            s = s * 1664525u + 1013904223u;
            const unsigned row2 = (s >> 16) % ROWS;
            d2 x = *reinterpret_cast<d2 *>(&lds[row * 128 + lane * 2]);
            __hip_atomic_fetch_add(&lds[row2 * 128 + lane * 2], x.x * 0.5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lds[row2 * 128 + lane * 2 + 1], x.y * 0.5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = lds[5] + acc0 + acc1;
}
template <int MODE> void run(const char *name, double bytes_per_iter) {
    double *d; hipMalloc(&d, 4096 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, 2u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double total = 256.0 * 16 * ITERS * bytes_per_iter;
    printf("%-28s %8.3f ms  %7.1f TB/s  %6.1f B/clk/CU @2.1GHz  (%.2f clk per wave-iter per CU)\n", name, ms, total / ms / 1e9,
           total / 256 / (ms * 1e-3 * 2.1e9), ms * 1e-3 * 2.1e9 / (16.0 * ITERS));
}
int main() {
    run<0>("2x ds_add_f64 (1 KB row)", 1024);
    run<1>("ds_read_b128 (1 KB row)", 1024);
    run<2>("read+write b128 RMW", 2048);
    run<3>("read b128 + 2 atomic add", 2048);
    return 0;
}
