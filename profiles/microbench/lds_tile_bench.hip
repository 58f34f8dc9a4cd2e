// Micro-benchmark for an LDS-tiled sparse x dense product with register accumulators selected through VGPR index
// mode (s_set_gpr_idx_on): does it beat the L2 row gather (gather_bench.hip: ~9.4 ns per nonzero per CU at l = 100)?
//   hipcc --offload-arch=gfx950 -O3 -o lds_tile_bench lds_tile_bench.hip && ./lds_tile_bench
// A workgroup of NW waves owns 16*NW outer vectors (16 per wave, 2 x 16 f64 accumulators per lane) and walks the
// panel in tiles of 128 rows staged in LDS; the nonzeros of one (16-vector group, tile) pair are a packed stream
// (4-bit vector, 7-bit row in tile, 21-bit count); each nonzero costs one ds_read_b128 and two v_fma_f64 whose
// destination is picked by the index register — no per-vector control flow.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d16 __attribute__((ext_vector_type(16)));
constexpr int TR = 128; // panel rows per tile

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ double bcastd(double v, uint32_t l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)l);
    return __hiloint2double(hi, lo);
}
// acc[r] += f * (x, y) for the wave-uniform r (r2 = 2 r: the index counts 32-bit registers)
__device__ __forceinline__ void acc_step(d16 &ax, d16 &ay, uint32_t r2, double f, d2 x) {
    asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC2,DST)\n\t"
                 "v_fma_f64 v[0:1], %3, %4, v[0:1]\n\t"
                 "v_fma_f64 v[32:33], %3, %5, v[32:33]\n\t"
                 "s_set_gpr_idx_off"
                 : "+{v[0:31]}"(ax), "+{v[32:63]}"(ay)
                 : "s"(r2), "s"(f), "v"(x.x), "v"(x.y)
                 : "m0");
}

// four nonzeros under ONE index-mode window: only M0[7:0] changes in between (s_set_gpr_idx_idx)
__device__ __forceinline__ void acc_step4(d16 &ax, d16 &ay, uint32_t ra, uint32_t rb, uint32_t rc, uint32_t rd, double fa, double fb,
                                          double fc, double fd, d2 xa, d2 xb, d2 xc, d2 xd) {
    asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC2,DST)\n\t"
                 "v_fma_f64 v[0:1], %6, %10, v[0:1]\n\t"
                 "v_fma_f64 v[32:33], %6, %11, v[32:33]\n\t"
                 "s_set_gpr_idx_idx %3\n\t"
                 "v_fma_f64 v[0:1], %7, %12, v[0:1]\n\t"
                 "v_fma_f64 v[32:33], %7, %13, v[32:33]\n\t"
                 "s_set_gpr_idx_idx %4\n\t"
                 "v_fma_f64 v[0:1], %8, %14, v[0:1]\n\t"
                 "v_fma_f64 v[32:33], %8, %15, v[32:33]\n\t"
                 "s_set_gpr_idx_idx %5\n\t"
                 "v_fma_f64 v[0:1], %9, %16, v[0:1]\n\t"
                 "v_fma_f64 v[32:33], %9, %17, v[32:33]\n\t"
                 "s_set_gpr_idx_off"
                 : "+{v[0:31]}"(ax), "+{v[32:63]}"(ay)
                 : "s"(ra), "s"(rb), "s"(rc), "s"(rd), "s"(fa), "s"(fb), "s"(fc), "s"(fd), "v"(xa.x), "v"(xa.y), "v"(xb.x), "v"(xb.y),
                   "v"(xc.x), "v"(xc.y), "v"(xd.x), "v"(xd.y)
                 : "m0");
}

__device__ __forceinline__ void acc_step_nop(d16 &ax, d16 &ay, uint32_t r2, double f, d2 x) {
    asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC2,DST)\n\t"
                 "s_nop 1\n\t"
                 "v_fma_f64 v[0:1], %3, %4, v[0:1]\n\t"
                 "v_fma_f64 v[32:33], %3, %5, v[32:33]\n\t"
                 "s_nop 1\n\t"
                 "s_set_gpr_idx_off"
                 : "+{v[0:31]}"(ax), "+{v[32:63]}"(ay)
                 : "s"(r2), "s"(f), "v"(x.x), "v"(x.y)
                 : "m0");
}

// MODE 0: stage every tile (single LDS buffer, serial); 1: stage tile 0 only (times the accumulate loop alone, results
// meaningless); 2: like 0 but the accumulator is picked by compiler-generated indexing (reference for the asm form);
// 3: like 0 with an s_nop after s_set_gpr_idx_on; 4: like 0, one index-mode window per 4 nonzeros; 5: 4 + accumulate only
template <int NW, int DEPTH, int MODE>
__global__ __launch_bounds__(64 * NW) void tile_kernel(const uint32_t *__restrict__ offs, const uint32_t *__restrict__ packed,
                                                       uint32_t n_tiles, const double *__restrict__ X, uint32_t ld, uint32_t l,
                                                       const double *__restrict__ sc_in, const double *__restrict__ sc_out,
                                                       double *__restrict__ out) {
    extern __shared__ d2 lds2[]; // TR x l panel tile, then TR inner scales
    double *lds = reinterpret_cast<double *>(lds2);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t group = blockIdx.x * NW + wave;
    const uint32_t col = lane * 2u;
    const bool act = col < l;
    const uint32_t lcol = act ? col : 0u;
    d16 ax = {0}, ay = {0};
    const double my_sc = sc_out[(size_t)group * 16u + (lane & 15u)];
    double *sc_t = lds + (size_t)TR * l;
    // offsets of 64 consecutive tiles sit one per lane (refreshed every 32 tiles); the first 64 packed nonzeros of
    // tile t+1 are fetched before tile t is worked on, so no global-load latency sits between two tiles
    const uint32_t *__restrict__ og = offs + (size_t)group * n_tiles;
    const size_t og_left = (size_t)(gridDim.x * NW - group) * n_tiles; // entries from og to the end of the table (inclusive end)
    uint32_t ov = lane <= og_left ? og[lane] : 0u;
    uint32_t pk_cur = 0;
    {
        const uint32_t a0 = rdlane(ov, 0), a1 = rdlane(ov, 1);
        if (a0 + lane < a1) pk_cur = packed[a0 + lane];
    }
    for (uint32_t t = 0; t < n_tiles; t++) {
        const uint32_t tb = t & ~31u;
        if (t == tb && t > 0) ov = (size_t)tb + lane <= og_left ? og[tb + lane] : 0u;
        const uint32_t o0 = rdlane(ov, t - tb), o1 = rdlane(ov, t - tb + 1);
        uint32_t pk_next = 0;
        if (t + 1 < n_tiles) {
            const uint32_t o2 = rdlane(ov, t - tb + 2);
            if (o1 + lane < o2) pk_next = packed[o1 + lane];
        }
        // stage the tile (contiguous TR*l doubles) + its scales
        const d2 *src = reinterpret_cast<const d2 *>(X + (size_t)t * TR * ld);
        d2 *dst = lds2;
        if ((MODE != 1 && MODE < 5) || t == 0) {
            for (uint32_t i = threadIdx.x; i < TR * l / 2u; i += 64u * NW) dst[i] = src[i];
            if (threadIdx.x < TR) sc_t[threadIdx.x] = sc_in[(size_t)t * TR + threadIdx.x];
        }
        __syncthreads();
        for (uint32_t c = o0; c < o1; c += 64u) {
            const uint32_t p = c + lane;
            uint32_t pk = pk_cur;
            if (c != o0) pk = p < o1 ? packed[p] : 0u;
            double f;
            {
                const uint32_t v = pk & 0x1fffffu, loc = (pk >> 21) & 127u, r = pk >> 28;
                const double so = __shfl(my_sc, (int)r, 64); // all lanes take part: a bpermute reads nothing from idle lanes
                f = p < o1 ? (MODE == 6 ? (1.0 + (double)v * sc_t[loc]) * so : log2(1.0 + (double)v * sc_t[loc]) * so) : 0.0;
            }
            const uint32_t n = min(64u, o1 - c);
            for (uint32_t j = 0; j < n; j += DEPTH) { // lanes past n hold pk = 0, f = 0: they add 0 * row 0 to vector 0
                d2 x[DEPTH];
                uint32_t r2[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    const uint32_t q = rdlane(pk, j + u);
                    r2[u] = (q >> 27) & 0x1eu;
                    const uint32_t loc = (q >> 21) & 127u;
                    if constexpr (MODE == 7) {
                        x[u] = (d2){(double)loc, (double)lcol};
                    } else {
                        x[u] = lds2[loc * (l >> 1) + (lcol >> 1)];
                    }
                }
                if constexpr (MODE == 8) {
#pragma unroll
                    for (int u = 0; u < DEPTH; u++) {
                        ax[0] += x[u].x;
                        ay[0] += x[u].y * (double)r2[u];
                    }
                } else if constexpr (MODE >= 4) {
#pragma unroll
                    for (int u = 0; u < DEPTH; u += 4)
                        acc_step4(ax, ay, r2[u], r2[u + 1], r2[u + 2], r2[u + 3], bcastd(f, j + u), bcastd(f, j + u + 1),
                                  bcastd(f, j + u + 2), bcastd(f, j + u + 3), x[u], x[u + 1], x[u + 2], x[u + 3]);
                } else
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    if constexpr (MODE == 2) {
                        const double fv = bcastd(f, j + u);
                        ax[r2[u] >> 1] = fma(fv, x[u].x, ax[r2[u] >> 1]);
                        ay[r2[u] >> 1] = fma(fv, x[u].y, ay[r2[u] >> 1]);
                    } else if constexpr (MODE == 3) {
                        acc_step_nop(ax, ay, r2[u], bcastd(f, j + u), x[u]);
                    } else {
                        acc_step(ax, ay, r2[u], bcastd(f, j + u), x[u]);
                    }
                }
            }
        }
        pk_cur = pk_next;
        __syncthreads();
    }
    if (act) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            d2 o = {ax[r], ay[r]};
            *reinterpret_cast<d2 *>(out + ((size_t)group * 16u + r) * l + col) = o;
        }
    }
}

// Second form: the per-nonzero scalars travel through LDS instead of v_readlane + SALU decoding. Once per 64 nonzeros the
// lanes write (weight f64, byte offset of the panel row in the tile, 2*vector) as 16 bytes to a per-wave strip; then
// every nonzero is: one broadcast ds_read_b128 of its strip entry, v_readfirstlane for the index register, one
// ds_read_b128 of the panel row, two indexed v_fma_f64 with the weight in a VGPR.  STAGED = 0 times the accumulate loop
// alone (tile 0 staged once).
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void acc_stepv(d16 &ax, d16 &ay, uint32_t r2, double f, d2 x) {
    asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC2,DST)\n\t"
                 "v_fma_f64 v[0:1], %3, %4, v[0:1]\n\t"
                 "v_fma_f64 v[32:33], %3, %5, v[32:33]\n\t"
                 "s_set_gpr_idx_off"
                 : "+{v[0:31]}"(ax), "+{v[32:63]}"(ay)
                 : "s"(r2), "v"(f), "v"(x.x), "v"(x.y)
                 : "m0");
}
template <int NW, int DEPTH, int STAGED>
__global__ __launch_bounds__(64 * NW) void tile_kernel2(const uint32_t *__restrict__ offs, const uint32_t *__restrict__ packed,
                                                        uint32_t n_tiles, const double *__restrict__ X, uint32_t ld, uint32_t l,
                                                        const double *__restrict__ sc_in, const double *__restrict__ sc_out,
                                                        double *__restrict__ out) {
    extern __shared__ d2 lds2[]; // TR x l panel tile | TR inner scales | NW strips of 64 x 16 B
    double *lds = reinterpret_cast<double *>(lds2);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t group = blockIdx.x * NW + wave;
    const uint32_t col = lane * 2u;
    const bool act = col < l;
    const uint32_t lcol = act ? col : 0u;
    d16 ax = {0}, ay = {0};
    const double my_sc = sc_out[(size_t)group * 16u + (lane & 15u)];
    double *sc_t = lds + (size_t)TR * l;
    u4 *strip = reinterpret_cast<u4 *>(lds + (size_t)TR * l + TR) + wave * 64u;
    const uint32_t *__restrict__ og = offs + (size_t)group * n_tiles;
    const size_t og_left = (size_t)(gridDim.x * NW - group) * n_tiles;
    uint32_t ov = lane <= og_left ? og[lane] : 0u;
    uint32_t pk_cur = 0;
    {
        const uint32_t a0 = rdlane(ov, 0), a1 = rdlane(ov, 1);
        if (a0 + lane < a1) pk_cur = packed[a0 + lane];
    }
    const char *ldsb = reinterpret_cast<const char *>(lds2) + lcol * 8u;
    for (uint32_t t = 0; t < n_tiles; t++) {
        const uint32_t tb = t & ~31u;
        if (t == tb && t > 0) ov = (size_t)tb + lane <= og_left ? og[tb + lane] : 0u;
        const uint32_t o0 = rdlane(ov, t - tb), o1 = rdlane(ov, t - tb + 1);
        uint32_t pk_next = 0;
        if (t + 1 < n_tiles) {
            const uint32_t o2 = rdlane(ov, t - tb + 2);
            if (o1 + lane < o2) pk_next = packed[o1 + lane];
        }
        if (STAGED || t == 0) {
            const d2 *src = reinterpret_cast<const d2 *>(X + (size_t)t * TR * ld);
            for (uint32_t i = threadIdx.x; i < TR * l / 2u; i += 64u * NW) lds2[i] = src[i];
            if (threadIdx.x < TR) sc_t[threadIdx.x] = sc_in[(size_t)t * TR + threadIdx.x];
        }
        __syncthreads();
        for (uint32_t c = o0; c < o1; c += 64u) {
            const uint32_t p = c + lane;
            uint32_t pk = pk_cur;
            if (c != o0) pk = p < o1 ? packed[p] : 0u;
            {
                const uint32_t v = pk & 0x1fffffu, loc = (pk >> 21) & 127u, r = pk >> 28;
                const double so = __shfl(my_sc, (int)r, 64);
                const double f = p < o1 ? log2(1.0 + (double)v * sc_t[loc]) * so : 0.0;
                u4 m;
                m.x = (uint32_t)__double2loint(f);
                m.y = (uint32_t)__double2hiint(f);
                m.z = loc * l * 8u;
                m.w = r * 2u;
                strip[lane] = m;
            }
            const uint32_t n = min(64u, o1 - c);
            for (uint32_t j = 0; j < n; j += DEPTH) {
                u4 m[DEPTH];
                d2 x[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; u++) m[u] = strip[j + u];
#pragma unroll
                for (int u = 0; u < DEPTH; u++) x[u] = *reinterpret_cast<const d2 *>(ldsb + m[u].z);
#pragma unroll
                for (int u = 0; u < DEPTH; u++)
                    acc_stepv(ax, ay, (uint32_t)__builtin_amdgcn_readfirstlane((int)m[u].w), __hiloint2double((int)m[u].y, (int)m[u].x), x[u]);
            }
        }
        pk_cur = pk_next;
        __syncthreads();
    }
    if (act) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            d2 o = {ax[r], ay[r]};
            *reinterpret_cast<d2 *>(out + ((size_t)group * 16u + r) * l + col) = o;
        }
    }
}

template <int NW, int DEPTH, int STAGED>
static void run2(const char *name, const uint32_t *offs, const uint32_t *packed, uint32_t n_groups, uint32_t n_tiles, const double *X,
                 uint32_t l, const double *sc_in, const double *sc_out, double *out, double nnz, int n_cu, std::vector<double> *host_out) {
    const size_t shmem = ((size_t)TR * l + TR) * 8 + (size_t)NW * 64 * 16;
    CK(hipFuncSetAttribute((const void *)tile_kernel2<NW, DEPTH, STAGED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const dim3 grid(n_groups / NW), block(64 * NW);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((tile_kernel2<NW, DEPTH, STAGED>), grid, block, shmem, 0, offs, packed, n_tiles, X, l, l, sc_in, sc_out, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 3;
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL((tile_kernel2<NW, DEPTH, STAGED>), grid, block, shmem, 0, offs, packed, n_tiles, X, l, l, sc_in, sc_out, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%-28s l=%3u  %8.3f ms  %6.2f ns/nnz/CU  (grid %u x %d threads)\n", name, l, ms, ms * 1e6 * n_cu / nnz, grid.x, 64 * NW);
    fflush(stdout);
    if (host_out) {
        host_out->resize((size_t)n_groups * 16 * l);
        CK(hipMemcpy(host_out->data(), out, host_out->size() * 8, hipMemcpyDeviceToHost));
    }
}

template <int NW, int DEPTH, int MODE>
static void run(const char *name, const uint32_t *offs, const uint32_t *packed, uint32_t n_groups, uint32_t n_tiles, const double *X,
                uint32_t l, const double *sc_in, const double *sc_out, double *out, double nnz, int n_cu, std::vector<double> *host_out) {
    const size_t shmem = ((size_t)TR * l + TR) * 8;
    CK(hipFuncSetAttribute((const void *)tile_kernel<NW, DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const dim3 grid(n_groups / NW), block(64 * NW);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((tile_kernel<NW, DEPTH, MODE>), grid, block, shmem, 0, offs, packed, n_tiles, X, l, l, sc_in, sc_out, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 3;
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL((tile_kernel<NW, DEPTH, MODE>), grid, block, shmem, 0, offs, packed, n_tiles, X, l, l, sc_in, sc_out, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%-28s l=%3u  %8.3f ms  %6.2f ns/nnz/CU  (grid %u x %d threads)\n", name, l, ms, ms * 1e6 * n_cu / nnz, grid.x, 64 * NW);
    fflush(stdout);
    if (host_out) {
        host_out->resize((size_t)n_groups * 16 * l);
        CK(hipMemcpy(host_out->data(), out, host_out->size() * 8, hipMemcpyDeviceToHost));
    }
}

int main(int argc, char **argv) {
    const uint32_t l = 100;
    const uint32_t n_tiles = argc > 1 ? (uint32_t)atoi(argv[1]) : 258u;  // 33 k panel rows
    const uint32_t n_groups = argc > 2 ? (uint32_t)atoi(argv[2]) : 8192u; // 131072 outer vectors
    const double density = 0.03;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::vector<uint32_t> offs((size_t)n_groups * n_tiles + 1);
    std::vector<uint32_t> packed;
    packed.reserve((size_t)(n_groups * (double)n_tiles * 16 * TR * density * 1.05));
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return s;
    };
    auto unif = [&]() { return ((double)(rnd() >> 11) + 0.5) * (1.0 / 9007199254740992.0); };
    const double inv_log = 1.0 / std::log(1.0 - density);
    for (uint32_t g = 0; g < n_groups; g++)
        for (uint32_t t = 0; t < n_tiles; t++) {
            offs[(size_t)g * n_tiles + t] = (uint32_t)packed.size();
            // geometric skipping over the 16 x TR cells of the (group, tile) block, vector-major
            for (double cell = std::floor(std::log(unif()) * inv_log); cell < 16.0 * TR;
                 cell += 1.0 + std::floor(std::log(unif()) * inv_log)) {
                const uint32_t ce = (uint32_t)cell, r = ce / TR, loc = ce % TR, v = 1u + (uint32_t)(rnd() % 7u);
                packed.push_back((r << 28) | (loc << 21) | v);
            }
        }
    offs.back() = (uint32_t)packed.size();
    const double nnz = (double)packed.size();
    printf("device %s, %d CUs; %u groups x %u tiles, nnz %.0f (%.1f per group-tile)\n", prop.name, n_cu, n_groups, n_tiles, nnz,
           nnz / ((double)n_groups * n_tiles));
    std::vector<double> hX((size_t)n_tiles * TR * l), hin((size_t)n_tiles * TR), hout((size_t)n_groups * 16);
    for (auto &x : hX) x = (double)(rnd() >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    for (auto &x : hin) x = 0.5 + (double)(rnd() >> 11) * (1.0 / 9007199254740992.0);
    for (auto &x : hout) x = 0.5 + (double)(rnd() >> 11) * (1.0 / 9007199254740992.0);
    uint32_t *doffs, *dpk;
    double *dX, *din, *dout, *dres;
    CK(hipMalloc(&doffs, offs.size() * 4));
    CK(hipMalloc(&dpk, packed.size() * 4));
    CK(hipMalloc(&dX, hX.size() * 8));
    CK(hipMalloc(&din, hin.size() * 8));
    CK(hipMalloc(&dout, hout.size() * 8));
    CK(hipMalloc(&dres, (size_t)n_groups * 16 * l * 8));
    CK(hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpk, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(din, hin.data(), hin.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dout, hout.data(), hout.size() * 8, hipMemcpyHostToDevice));
    auto check = [&](const std::vector<double> &res, const char *what) {
        double worst = 0.0;
        for (uint32_t g : {0u, 1u, n_groups / 2u, n_groups - 1u})
            for (uint32_t r = 0; r < 16; r += 5) {
                std::vector<double> ref(l, 0.0);
                for (uint32_t t = 0; t < n_tiles; t++)
                    for (uint32_t p = offs[(size_t)g * n_tiles + t]; p < offs[(size_t)g * n_tiles + t + 1]; p++) {
                        const uint32_t pk = packed[p];
                        if ((pk >> 28) != r) continue;
                        const uint32_t loc = (pk >> 21) & 127u, v = pk & 0x1fffffu;
                        const double f = std::log2(1.0 + (double)v * hin[(size_t)t * TR + loc]) * hout[(size_t)g * 16 + r];
                        for (uint32_t c = 0; c < l; c++) ref[c] += f * hX[((size_t)t * TR + loc) * l + c];
                    }
                double w2 = 0.0;
                for (uint32_t c = 0; c < l; c++)
                    w2 = std::max(w2, std::abs(ref[c] - res[((size_t)g * 16 + r) * l + c]) / (1.0 + std::abs(ref[c])));
                worst = std::max(worst, w2);
            }
        printf("   %s: max rel deviation from the host evaluation %.3e\n", what, worst);
    };
    std::vector<double> res;
    run<16, 8, 2>("16w d8 compiler-indexed", doffs, dpk, n_groups, n_tiles, dX, l, din, dout, dres, nnz, n_cu, &res);
    check(res, "compiler-indexed");
    run<16, 8, 0>("16w d8 asm-indexed", doffs, dpk, n_groups, n_tiles, dX, l, din, dout, dres, nnz, n_cu, &res);
    check(res, "asm-indexed");
    run<16, 8, 5>("16w d8 accumulate only", doffs, dpk, n_groups, n_tiles, dX, l, din, dout, dres, nnz, n_cu, nullptr);
    run2<16, 4, 1>("strip 16w d4 staged", doffs, dpk, n_groups, n_tiles, dX, l, din, dout, dres, nnz, n_cu, &res);
    check(res, "strip form");
    run2<16, 4, 0>("strip 16w d4 acc only", doffs, dpk, n_groups, n_tiles, dX, l, din, dout, dres, nnz, n_cu, nullptr);
    run2<16, 2, 0>("strip 16w d2 acc only", doffs, dpk, n_groups, n_tiles, dX, l, din, dout, dres, nnz, n_cu, nullptr);
    run2<16, 8, 0>("strip 16w d8 acc only", doffs, dpk, n_groups, n_tiles, dX, l, din, dout, dres, nnz, n_cu, nullptr);
    run2<8, 4, 0>("strip 8w d4 acc only", doffs, dpk, n_groups, n_tiles, dX, l, din, dout, dres, nnz, n_cu, nullptr);
    run2<8, 8, 0>("strip 8w d8 acc only", doffs, dpk, n_groups, n_tiles, dX, l, din, dout, dres, nnz, n_cu, nullptr);
    return 0;
}
