// Micro-benchmark behind the row-gather design of spmm_gather2d_kernel: what bounds a wave that
// gathers 800-byte panel rows from an L2-resident slice — bytes, load instructions, or latency?
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip && ./gather_bench
// Every variant: one wave per "outer vector" of LEN nonzeros (u32 index + f64 weight, read coalesced 64 at a time),
// each nonzero gathers one row of the slice (ROWS x L doubles) and accumulates weight * row into 2 columns per lane.
// Prints ns per nonzero per CU and the gathered TB/s.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e = (x);                                                                     \
        if (e != hipSuccess) {                                                                  \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));            \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ double bcastd(double v, uint32_t l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)l);
    return __hiloint2double(hi, lo);
}

// MODE 0: flat loads, idle lanes re-read column pair 0 (the shipped kernel)
// MODE 1: flat loads, idle lanes masked off with exec for the whole gather loop
// MODE 2: buffer loads (SGPR row offset + constant VGPR lane offset), idle lanes re-read pair 0
// MODE 3: buffer loads, idle lanes masked off
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t *__restrict__ ind, const double *__restrict__ wgt, uint32_t len,
                                                     uint64_t n_outer, const double *__restrict__ X, uint32_t ld, uint32_t l,
                                                     double *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t row = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= n_outer) return;
    const uint32_t *__restrict__ ip = ind + row * len;
    const double *__restrict__ wp = wgt + row * len;
    const uint32_t col = lane * 2u;
    const bool act = col < l;
    const uint32_t lcol = act ? col : 0u;
    d2 acc = {0.0, 0.0};
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, 0x7fffffff, 0x00020000);
    for (uint32_t c = 0; c < len; c += 64u) {
        const uint32_t p = c + lane;
        uint32_t idx = 0;
        double f = 0.0;
        if (p < len) {
            idx = ip[p];
            f = wp[p];
        }
        const uint32_t n = min(64u, len - c);
        auto body = [&]() {
            uint32_t j = 0;
            for (; j + DEPTH <= n; j += DEPTH) {
                d2 x[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    const uint32_t g = rdlane(idx, j + u);
                    if constexpr (MODE >= 2) {
                        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lcol * 8u, g * ld * 8u, 0);
                        x[u].x = __hiloint2double((int)v.y, (int)v.x);
                        x[u].y = __hiloint2double((int)v.w, (int)v.z);
                    } else {
                        x[u] = *reinterpret_cast<const d2 *>(X + (size_t)g * ld + lcol);
                    }
                }
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    const double fv = bcastd(f, j + u);
                    acc.x = fma(fv, x[u].x, acc.x);
                    acc.y = fma(fv, x[u].y, acc.y);
                }
            }
            for (; j < n; j++) {
                const uint32_t g = rdlane(idx, j);
                const double fv = bcastd(f, j);
                d2 x;
                if constexpr (MODE >= 2) {
                    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lcol * 8u, g * ld * 8u, 0);
                    x.x = __hiloint2double((int)v.y, (int)v.x);
                    x.y = __hiloint2double((int)v.w, (int)v.z);
                } else {
                    x = *reinterpret_cast<const d2 *>(X + (size_t)g * ld + lcol);
                }
                acc.x = fma(fv, x.x, acc.x);
                acc.y = fma(fv, x.y, acc.y);
            }
        };
        if constexpr (MODE == 1 || MODE == 3) {
            if (act) body();
        } else {
            body();
        }
    }
    if (act) *reinterpret_cast<d2 *>(out + row * ld + col) = acc;
}

template <int MODE, int DEPTH>
static void run(const char *name, const uint32_t *ind, const double *wgt, uint32_t len, uint64_t n_outer, const double *X, uint32_t ld,
                uint32_t l, double *out, int n_cu) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const dim3 grid((unsigned)((n_outer + 3) / 4)), block(256);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((gather_kernel<MODE, DEPTH>), grid, block, 0, 0, ind, wgt, len, n_outer, X, ld, l, out);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((gather_kernel<MODE, DEPTH>), grid, block, 0, 0, ind, wgt, len, n_outer, X, ld, l, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double nnz = (double)n_outer * len;
    printf("%-34s l=%3u len=%4u  %8.3f ms  %6.2f ns/nnz/CU  %6.2f TB/s gathered\n", name, l, len, ms, ms * 1e6 * n_cu / nnz,
           nnz * l * 8.0 / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const uint32_t rows = argc > 1 ? (uint32_t)atoi(argv[1]) : 3750u; // slice rows (3 MB at l = 100)
    const uint64_t n_outer = argc > 2 ? (uint64_t)atoll(argv[2]) : 1000000ull;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, slice rows %u\n", prop.name, n_cu, rows);
    const uint32_t len_a = argc > 3 ? (uint32_t)atoi(argv[3]) : 112u, len_b = argc > 4 ? (uint32_t)atoi(argv[4]) : 1024u;
    for (uint32_t l : {100u, 128u, 64u}) {
        if (argc > 3 && l != 100u) continue;
        for (uint32_t len : {len_a, len_b}) {
            const uint64_t no = len <= 128u ? n_outer : n_outer / 8;
            const uint32_t ld = l;
            std::vector<uint32_t> hi(no * len);
            std::vector<double> hw(no * len);
            uint64_t s = 88172645463325252ull;
            for (size_t i = 0; i < hi.size(); i++) {
                s ^= s << 13;
                s ^= s >> 7;
                s ^= s << 17;
                hi[i] = (uint32_t)(s % rows);
                hw[i] = 1.0 + (double)(s >> 40) * 1e-9;
            }
            // ascending within an outer vector, like the stored indices
            for (uint64_t r = 0; r < no; r++) std::sort(hi.begin() + r * len, hi.begin() + (r + 1) * len);
            uint32_t *di;
            double *dw, *dX, *dout;
            CK(hipMalloc(&di, hi.size() * 4));
            CK(hipMalloc(&dw, hw.size() * 8));
            CK(hipMalloc(&dX, (size_t)rows * ld * 8));
            CK(hipMalloc(&dout, no * ld * 8));
            CK(hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dw, hw.data(), hw.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemset(dX, 0, (size_t)rows * ld * 8));
            run<0, 8>("flat, idle->pair0, depth 8", di, dw, len, no, dX, ld, l, dout, n_cu);
            run<1, 8>("flat, idle masked, depth 8", di, dw, len, no, dX, ld, l, dout, n_cu);
            run<2, 8>("buffer, idle->pair0, depth 8", di, dw, len, no, dX, ld, l, dout, n_cu);
            run<3, 8>("buffer, idle masked, depth 8", di, dw, len, no, dX, ld, l, dout, n_cu);
            run<2, 16>("buffer, idle->pair0, depth 16", di, dw, len, no, dX, ld, l, dout, n_cu);
            run<3, 16>("buffer, idle masked, depth 16", di, dw, len, no, dX, ld, l, dout, n_cu);
            run<0, 16>("flat, idle->pair0, depth 16", di, dw, len, no, dX, ld, l, dout, n_cu);
            run<3, 4>("buffer, idle masked, depth 4", di, dw, len, no, dX, ld, l, dout, n_cu);
            CK(hipFree(di));
            CK(hipFree(dw));
            CK(hipFree(dX));
            CK(hipFree(dout));
        }
    }
    return 0;
}
