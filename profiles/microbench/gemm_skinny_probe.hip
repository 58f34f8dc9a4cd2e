// Probe of the LDS-free narrow product (scan-rs_amd/csrc/dense_skinny.inc) at the Ritz factor's shape: X 10^6 x 500 (ld 500),
// W 500 x 50 -> Out 10^6 x 50. Checks every entry against a plain one-thread-per-entry FMA loop and times the whole product and
// the same in 8 row blocks (the delivery pipeline's form). Build: hipcc -O3 --offload-arch=gfx950 gemm_skinny_probe.hip -o gemm_skinny_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
#include "../../scan-rs_amd/csrc/dense_skinny.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill_kernel(double *p, uint64_t n, uint64_t seed) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (double)(int64_t)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}
__global__ void ref_kernel(const double *X, uint32_t ldx, uint32_t n, const double *W, uint32_t ldw, uint32_t m, uint64_t rows, double *Out, uint32_t ldo) {
    const uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (e >= rows * m) return;
    const uint64_t r = e / m;
    const uint32_t j = (uint32_t)(e - r * m);
    double s = 0.0;
    for (uint32_t k = 0; k < n; k++) s = fma(X[r * ldx + k], W[(size_t)k * ldw + j], s);
    Out[r * ldo + j] = s;
}
template <int NT>
static void run(const double *X, uint32_t ldx, uint32_t n, const double *Wt, uint32_t n_pad, uint32_t m, uint64_t rows, double *Out, uint32_t ldo, hipStream_t s) {
    const uint64_t per_wg = 4ull * 16 * SKD_MT;
    hipLaunchKernelGGL(gemm_skinny_direct_kernel<NT>, dim3((unsigned)((rows + per_wg - 1) / per_wg), (m + 16 * NT - 1) / (16 * NT)), dim3(256), 0, s, X, ldx, n, Wt, n_pad, m, rows, 1.0, 0.0,
                       nullptr, 0, Out, ldo, nullptr);
}
int main(int argc, char **argv) {
    const uint64_t rows = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
    const uint32_t n = argc > 2 ? atoi(argv[2]) : 500, m = argc > 3 ? atoi(argv[3]) : 50;
    const uint32_t ldx = argc > 4 ? atoi(argv[4]) : ((n + 1) & ~1u), ldo = (m + 1) & ~1u, n_pad = (n + 15) / 16 * 16;
    const uint32_t gcols = argc > 5 ? atoi(argv[5]) : 64; // columns per group (multiple of 16, <= 112)
    const uint32_t groups = (m + gcols - 1) / gcols, nt = ((m + 15) / 16 + groups - 1) / groups, m_pad = nt * 16 * groups;
    double *X, *W, *Wt, *Out, *Ref;
    CK(hipMalloc(&X, rows * ldx * 8));
    CK(hipMalloc(&W, (size_t)n * m * 8));
    CK(hipMalloc(&Wt, (size_t)n_pad * m_pad * 8));
    CK(hipMalloc(&Out, rows * ldo * 8));
    CK(hipMalloc(&Ref, rows * ldo * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, rows * ldx, 1);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, W, (uint64_t)n * m, 77);
    CK(hipMemset(Out, 0, rows * ldo * 8));
    CK(hipMemset(Ref, 0, rows * ldo * 8));
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((rows * m + 255) / 256)), dim3(256), 0, 0, X, ldx, n, W, m, m, rows, Ref, ldo);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto product = [&](uint64_t r_begin, uint64_t nr) {
        const double *x = X + r_begin * ldx;
        double *o = Out + r_begin * ldo;
        switch (nt) {
        case 1: run<1>(x, ldx, n, Wt, n_pad, m, nr, o, ldo, 0); break;
        case 2: run<2>(x, ldx, n, Wt, n_pad, m, nr, o, ldo, 0); break;
        case 3: run<3>(x, ldx, n, Wt, n_pad, m, nr, o, ldo, 0); break;
        case 4: run<4>(x, ldx, n, Wt, n_pad, m, nr, o, ldo, 0); break;
        case 5: run<5>(x, ldx, n, Wt, n_pad, m, nr, o, ldo, 0); break;
        case 6: run<6>(x, ldx, n, Wt, n_pad, m, nr, o, ldo, 0); break;
        default: run<7>(x, ldx, n, Wt, n_pad, m, nr, o, ldo, 0); break;
        }
    };
    for (int form = 0; form < 2; form++) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(skinny_wt_kernel, dim3((n_pad * m_pad + 255) / 256), dim3(256), 0, 0, W, m, n, m, n_pad, m_pad, Wt, nullptr);
            if (form == 0)
                product(0, rows);
            else {
                const uint64_t blk = (((rows + 7) / 8) + 1023) & ~1023ull;
                for (uint64_t r = 0; r < rows; r += blk) product(r, std::min(blk, rows - r));
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("rows %llu n %u m %u MT %u %s: best %.3f ms = %.1f TFLOP/s (useful), %.2f TB/s of X\n", (unsigned long long)rows, n, m, SKD_MT,
               form ? "8 row blocks" : "one launch", best, 2.0 * rows * n * m / best * 1e-9, rows * (double)ldx * 8 / best * 1e-9);
    }
    std::vector<double> ho(rows * ldo), hr(rows * ldo);
    CK(hipMemcpy(ho.data(), Out, ho.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), Ref, hr.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (uint64_t r = 0; r < rows; r++)
        for (uint32_t j = 0; j < m; j++) worst = std::max(worst, std::fabs(ho[r * ldo + j] - hr[r * ldo + j]));
    printf("max |direct - reference| = %.3e (entries ~ sqrt(n)/12 in size) -> %s\n", worst, worst < 1e-12 ? "ok" : "MISMATCH");
    return worst < 1e-12 ? 0 : 1;
}
