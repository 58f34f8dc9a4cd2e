// Micro-benchmark: LDS-tiled sparse x dense product with STATIC register accumulators (round 2).
//   hipcc --offload-arch=gfx950 -O3 -o slot_tile_bench slot_tile_bench.hip && ./slot_tile_bench
//
// Round 1 (lds_tile_bench.hip) picked the accumulator of a nonzero through VGPR index mode and lost to the L2 row gather
// (7.6-9.9 ns vs 9.3 ns per nonzero per CU). Here no register is ever indexed dynamically:
//   * a wave owns S outer vectors ("slots") for the whole kernel: 2 f64 accumulators = 4 VGPRs per lane per slot (lane = column pair);
//   * a workgroup of NW waves walks the panel in tiles of TR rows staged in LDS (double-buffered, LDS-DMA);
//   * the nonzeros of one (wave, tile) visit are stored sorted by slot; the code is unrolled over the S slots, each slot a
//     little loop over ITS nonzeros of this tile (count from a packed descriptor), so the destination registers of the two
//     v_fma_f64 are compile-time constants; per nonzero: (byte offset, weight) by v_readlane from the lane-parallel decode of
//     the visit's records, one ds_read_b128 of the panel row, two v_fma_f64.
// Slots come in sets of SS whose records (<= 64 per set and visit, guaranteed by the builder) are decoded / mapped one
// per lane (one log2 per lane) right before the set's serial part.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
constexpr int DESC_DW = 32; // dwords per (group, tile) descriptor: [0, 24) packed u8 counts, [24, 32) first record of each set

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ double bcastd(double v, uint32_t l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)l);
    return __hiloint2double(hi, lo);
}

typedef __attribute__((address_space(3))) void *lds_ptr_t;

// the product's map logarithm (kernels.hip log_core / map_log2: fdlibm kernel, ~45 VALU instructions, few registers)
__device__ __forceinline__ double fast_log2(double x) {
    int e;
    double m = frexp(x, &e);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double kd = (double)e;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double s = f * r;
    s = fma(fma(-d, s, f), r, s);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 =
        z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double lm = f - (hfsq - s * (hfsq + R));
    return fma(lm, 1.44269504088896338700e+00, kd);
}

// MODE bit 0: the last wave of the workgroup only stages tiles (a "loader"), the others only compute
// MODE bit 1: stage tile 0 only (times the accumulate structure alone; results meaningless)
// MODE bit 2: no log2 in the lane-parallel decode (weight = count * scale)
// MODE bit 3: every wave issues its share of the next tile's LDS-DMA right after the barrier (default: before the last set's serial part)
template <int S, int SS, int NW, int TR, int G, int MODE>
__global__ __launch_bounds__(64 * NW) void slot_kernel(const uint32_t *__restrict__ desc, const uint16_t *__restrict__ key,
                                                       const uint32_t *__restrict__ val, uint32_t n_tiles,
                                                       const double *__restrict__ X, uint32_t ld, uint32_t l,
                                                       const double *__restrict__ sc_out, double *__restrict__ out) {
    constexpr int NSET = S / SS;
    static_assert(S % SS == 0 && NSET <= 8 && S <= 96, "descriptor layout");
    constexpr bool LOADER = (MODE & 1) != 0;
    constexpr int NC = LOADER ? NW - 1 : NW; // computing waves
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t rowbytes = ld * 8u;
    const uint32_t tile_bytes = TR * rowbytes;
    const uint32_t n_chunks = tile_bytes / 1024u; // builder guarantees divisibility
    const bool is_loader = LOADER && wave == NW - 1;
    const uint32_t stage_id = LOADER ? 0u : wave, stage_n = LOADER ? 1u : (uint32_t)NW;

    auto stage = [&](uint32_t t, uint32_t buf) {
        const char *srcu = reinterpret_cast<const char *>(X + (size_t)t * TR * ld); // wave-uniform
        char *dst = lds + buf * tile_bytes;
        const uint32_t l16 = lane * 16u;
        if (LOADER) {
            for (uint32_t c = 0; c < n_chunks; c++)
                __builtin_amdgcn_global_load_lds(srcu + c * 1024u + l16, (lds_ptr_t)(dst + c * 1024u), 16, 0, 0);
        } else {
            constexpr uint32_t CH = (TR * 1024u / 1024u + NW - 1) / NW; // chunks per wave at <= 1 KB rows
#pragma unroll
            for (uint32_t i = 0; i < CH; i++) {
                const uint32_t c = stage_id + i * stage_n;
                if (c < n_chunks) __builtin_amdgcn_global_load_lds(srcu + c * 1024u + l16, (lds_ptr_t)(dst + c * 1024u), 16, 0, 0);
            }
        }
    };

    if (is_loader || !LOADER) stage(0, 0);
    if (is_loader) {
        for (uint32_t t = 0; t < n_tiles; t++) {
            __syncthreads(); // tile t landed (vmcnt(0) by the fence), buffer (t+1)&1 free
            if (!(MODE & 2) && t + 1 < n_tiles) stage(t + 1, (t + 1) & 1u);
        }
        __syncthreads();
        return;
    }

    const uint32_t group = blockIdx.x * NC + wave;
    // the wave's S outer scales live in LDS behind the two tile buffers (one ds_read_b64 gather per record instead of a
    // dependent global load in the decode)
    double *wsc = reinterpret_cast<double *>(lds + 2u * tile_bytes) + wave * S;
    if (lane < S) wsc[lane] = sc_out[(size_t)group * S + lane];
    const uint32_t lcol16 = (lane * 2u < l ? lane : 0u) * 16u;
    d2 acc[S];
#pragma unroll
    for (int s = 0; s < S; s++) acc[s] = (d2){0.0, 0.0};

    const uint32_t *__restrict__ dg = desc + (size_t)group * n_tiles * DESC_DW;
    uint32_t vdesc = lane < DESC_DW ? dg[lane] : 0u;
    uint32_t rk[NSET], rv[NSET];
#pragma unroll
    for (int b = 0; b < NSET; b++) {
        const uint32_t st = rdlane(vdesc, 24 + b);
        rk[b] = key[st + lane];
        rv[b] = val[st + lane];
    }
    uint32_t vdesc_nxt = (n_tiles > 1 && lane < DESC_DW) ? dg[DESC_DW + lane] : 0u;

    for (uint32_t t = 0; t < n_tiles; t++) {
        __syncthreads(); // tile t is in buffer t&1
        if ((MODE & 8) && !LOADER && !(MODE & 2) && t + 1 < n_tiles) stage(t + 1, (t + 1) & 1u);
        const char *tile = lds + ((MODE & 2) ? 0u : (t & 1u) * tile_bytes) + lcol16;
        uint32_t vdesc_n2 = 0u;
#pragma unroll
        for (int b = 0; b < NSET; b++) {
            // lane-parallel decode + map of the set's records (lanes past the set's count hold junk that is never read)
            const uint32_t k = rk[b], v = rv[b];
            const uint32_t slot = min(k >> 8, (uint32_t)(S - 1)), row = k & 255u;
            const double so = wsc[slot];
            const double f = (MODE & 4) ? (double)v * so : fast_log2(1.0 + (double)v * so);
            const uint32_t voff = row * rowbytes;
            const uint32_t flo = (uint32_t)__double2loint(f), fhi = (uint32_t)__double2hiint(f);
            // refill the raw registers of this set with the next visit's records
            if (t + 1 < n_tiles) {
                const uint32_t st = rdlane(vdesc_nxt, 24 + b);
                rk[b] = key[st + lane];
                rv[b] = val[st + lane];
            }
            if (b == 0) vdesc_n2 = dg[(size_t)min(t + 2, n_tiles - 1) * DESC_DW + (lane & 31u)];
            if (b == NSET - 1 && !(MODE & 8) && !LOADER && !(MODE & 2) && t + 1 < n_tiles) stage(t + 1, (t + 1) & 1u);
            uint32_t p = 0;
#pragma unroll
            for (int q = 0; q < SS; q++) {
                const int s = b * SS + q;
                uint32_t n = (rdlane(vdesc, s >> 2) >> ((s & 3) * 8)) & 255u;
                while (n) {
                    const uint32_t m = min(n, (uint32_t)G);
                    d2 x[G];
                    double w[G];
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        const uint32_t pl = p + min((uint32_t)u, m - 1u);
                        const uint32_t off = rdlane(voff, pl);
                        const double wv = __hiloint2double((int)rdlane(fhi, pl), (int)rdlane(flo, pl));
                        w[u] = (uint32_t)u < m ? wv : 0.0;
                        x[u] = *reinterpret_cast<const d2 *>(tile + off);
                    }
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        acc[s].x = fma(w[u], x[u].x, acc[s].x);
                        acc[s].y = fma(w[u], x[u].y, acc[s].y);
                    }
                    p += m;
                    n -= m;
                }
            }
        }
        vdesc = vdesc_nxt;
        vdesc_nxt = vdesc_n2;
    }
    __syncthreads();
    if (lane * 2u < l) {
#pragma unroll
        for (int s = 0; s < S; s++) *reinterpret_cast<d2 *>(out + ((size_t)group * S + s) * ld + lane * 2u) = acc[s];
    }
}

// ------------------------------------------------------------------------------------------------------------------
struct Problem {
    uint32_t n_groups, n_tiles, S, SS, TR, ld, l;
    std::vector<uint32_t> desc;
    std::vector<uint16_t> key;
    std::vector<uint32_t> val;
    std::vector<double> X, sc;
    uint64_t nnz = 0;
};

static inline uint64_t xs(uint64_t &s) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
}

static void build(Problem &P, double density) {
    const uint32_t NSET = P.S / P.SS;
    P.desc.assign((size_t)P.n_groups * P.n_tiles * DESC_DW, 0u);
    P.key.clear();
    P.val.clear();
    P.key.reserve((size_t)(P.n_groups * (double)P.n_tiles * P.S * P.TR * density * 1.05) + 1024);
    P.val.reserve(P.key.capacity());
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    const double inv_log1m = 1.0 / log(1.0 - density);
    uint64_t clipped = 0;
    for (uint32_t g = 0; g < P.n_groups; g++) {
        for (uint32_t t = 0; t < P.n_tiles; t++) {
            uint32_t *d = &P.desc[((size_t)g * P.n_tiles + t) * DESC_DW];
            uint8_t *cnt = reinterpret_cast<uint8_t *>(d);
            for (uint32_t b = 0; b < NSET; b++) {
                d[24 + b] = (uint32_t)P.key.size();
                uint32_t in_set = 0;
                for (uint32_t q = 0; q < P.SS; q++) {
                    const uint32_t s = b * P.SS + q;
                    uint32_t c = 0;
                    // geometric gaps between present rows (Bernoulli(density) per row)
                    for (double r = -1.0;;) {
                        const double u = ((double)(xs(rng) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
                        r += 1.0 + floor(log(u) * inv_log1m);
                        if (r >= (double)P.TR) break;
                        if (in_set == 64) {
                            clipped++;
                            continue;
                        }
                        P.key.push_back((uint16_t)((s << 8) | (uint32_t)r));
                        P.val.push_back(1u + (uint32_t)(xs(rng) % 7u));
                        c++;
                        in_set++;
                    }
                    cnt[s] = (uint8_t)c;
                }
            }
        }
    }
    P.nnz = P.key.size();
    for (int i = 0; i < 128; i++) {
        P.key.push_back(0);
        P.val.push_back(0);
    }
    if (clipped) printf("   (builder clipped %llu records: sets over 64)\n", (unsigned long long)clipped);
    P.X.resize((size_t)P.n_tiles * P.TR * P.ld);
    for (auto &x : P.X) x = (double)(xs(rng) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    P.sc.resize((size_t)P.n_groups * P.S);
    for (auto &x : P.sc) x = 0.5 + (double)(xs(rng) >> 11) * (1.0 / 9007199254740992.0);
}

// host evaluation of one outer vector (group g, slot s)
static void host_row(const Problem &P, uint32_t g, uint32_t s, bool use_log, std::vector<double> &r) {
    r.assign(P.l, 0.0);
    const uint32_t NSET = P.S / P.SS;
    (void)NSET;
    for (uint32_t t = 0; t < P.n_tiles; t++) {
        const uint32_t *d = &P.desc[((size_t)g * P.n_tiles + t) * DESC_DW];
        const uint8_t *cnt = reinterpret_cast<const uint8_t *>(d);
        const uint32_t b = s / P.SS;
        uint32_t pos = d[24 + b];
        for (uint32_t q = b * P.SS; q < s; q++) pos += cnt[q];
        for (uint32_t j = 0; j < cnt[s]; j++) {
            const uint32_t k = P.key[pos + j], v = P.val[pos + j];
            const uint32_t row = k & 255u;
            const double so = P.sc[(size_t)g * P.S + s];
            const double f = use_log ? log2(1.0 + (double)v * so) : (double)v * so;
            const double *xr = &P.X[((size_t)t * P.TR + row) * P.ld];
            for (uint32_t c = 0; c < P.l; c++) r[c] = fma(f, xr[c], r[c]);
        }
    }
}

struct Dev {
    uint32_t *desc;
    uint16_t *key;
    uint32_t *val;
    double *X, *sc, *out;
};

template <int S, int SS, int NW, int TR, int G, int MODE>
static void run(const char *name, const Problem &P, const Dev &D, int n_cu, bool check) {
    constexpr bool LOADER = (MODE & 1) != 0;
    constexpr int NC = LOADER ? NW - 1 : NW;
    const size_t shmem = (size_t)2 * TR * P.ld * 8 + (size_t)NW * S * 8;
    auto kern = slot_kernel<S, SS, NW, TR, G, MODE>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const uint32_t n_wg = P.n_groups / NC;
    const dim3 grid(n_wg), block(64 * NW);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, grid, block, shmem, 0, D.desc, D.key, D.val, P.n_tiles, D.X, P.ld, P.l, D.sc, D.out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 3;
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(kern, grid, block, shmem, 0, D.desc, D.key, D.val, P.n_tiles, D.X, P.ld, P.l, D.sc, D.out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double nnz_used = (double)P.nnz * ((double)n_wg * NC / P.n_groups);
    printf("%-34s S=%2d SS=%2d NW=%2d TR=%3d G=%d  %8.3f ms  %6.2f ns/nnz/CU  (grid %u x %d, lds %zu KB)\n", name, S, SS, NW, TR, G, ms,
           ms * 1e6 * n_cu / nnz_used, grid.x, 64 * NW, shmem >> 10);
    fflush(stdout);
    if (check && !(MODE & 2)) {
        std::vector<double> ho((size_t)P.n_groups * S * P.ld);
        CK(hipMemcpy(ho.data(), D.out, ho.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        std::vector<double> r;
        for (uint32_t i = 0; i < 48; i++) {
            const uint32_t g = (uint32_t)(((uint64_t)i * 2654435761u) % (n_wg * NC)), s = (i * 7u) % S;
            host_row(P, g, s, !(MODE & 4), r);
            for (uint32_t c = 0; c < P.l; c++) {
                const double dv = ho[((size_t)g * S + s) * P.ld + c];
                worst = std::max(worst, fabs(dv - r[c]) / (1e-30 + fabs(r[c]) + 1e-9));
            }
        }
        printf("   max rel deviation from the host evaluation %.3e\n", worst);
    }
}

template <int S, int SS, int TR, int NW>
static void suite(int n_cu, uint32_t l, uint32_t wg_rounds) {
    // groups: enough for every NW tried; 33k inner positions; 3 % density
    Problem P;
    P.S = S;
    P.SS = SS;
    P.TR = TR;
    P.l = l;
    P.ld = l;
    P.n_tiles = (33000 + TR - 1) / TR;
    P.n_groups = (uint32_t)n_cu * wg_rounds * 16u;
    // keep the host build bounded: cap the tile count instead of the group count
    const double est = (double)P.n_groups * P.n_tiles * S * TR * 0.03;
    if (est > 2.5e8) P.n_tiles = std::max(8u, (uint32_t)(2.5e8 / ((double)P.n_groups * S * TR * 0.03)));
    build(P, 0.03);
    printf("S=%d SS=%d TR=%d: %u groups x %u tiles, nnz %llu (%.1f per group-tile)\n", S, SS, TR, P.n_groups, P.n_tiles,
           (unsigned long long)P.nnz, (double)P.nnz / ((double)P.n_groups * P.n_tiles));
    Dev D;
    CK(hipMalloc(&D.desc, P.desc.size() * 4));
    CK(hipMalloc(&D.key, P.key.size() * 2));
    CK(hipMalloc(&D.val, P.val.size() * 4));
    CK(hipMalloc(&D.X, P.X.size() * 8 + 4096));
    CK(hipMalloc(&D.sc, P.sc.size() * 8));
    CK(hipMalloc(&D.out, (size_t)P.n_groups * S * P.ld * 8));
    CK(hipMemcpy(D.desc, P.desc.data(), P.desc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.key, P.key.data(), P.key.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.val, P.val.data(), P.val.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.X, P.X.data(), P.X.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.sc, P.sc.data(), P.sc.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(D.out, 0, (size_t)P.n_groups * S * P.ld * 8));

    run<S, SS, NW, TR, 1, 0>("all stage, G=1", P, D, n_cu, true);
    run<S, SS, NW, TR, 2, 0>("all stage, G=2", P, D, n_cu, true);
    run<S, SS, NW, TR, 4, 0>("all stage, G=4", P, D, n_cu, false);
    run<S, SS, NW, TR, 2, 2>("tile 0 only, G=2", P, D, n_cu, false);
    run<S, SS, NW, TR, 2, 6>("tile 0 only, no log, G=2", P, D, n_cu, false);
    run<S, SS, NW, TR, 2, 1>("loader wave, G=2", P, D, n_cu, true);
    run<S, SS, NW, TR, 2, 8>("all stage early, G=2", P, D, n_cu, true);
    CK(hipFree(D.desc));
    CK(hipFree(D.key));
    CK(hipFree(D.val));
    CK(hipFree(D.X));
    CK(hipFree(D.sc));
    CK(hipFree(D.out));
}

int main(int argc, char **argv) {
    int dev = 0, n_cu = 256;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const uint32_t l = argc > 1 ? (uint32_t)atoi(argv[1]) : 100u;
    printf("device %d, %d CUs, l = %u\n", dev, n_cu, l);
    suite<24, 12, 96, 16>(n_cu, l, 2);
    suite<20, 10, 96, 16>(n_cu, l, 2);
    suite<32, 16, 96, 12>(n_cu, l, 2);
    suite<28, 14, 96, 12>(n_cu, l, 2);
    suite<48, 16, 96, 8>(n_cu, l, 2);
    suite<42, 14, 96, 8>(n_cu, l, 2);
    return 0;
}
