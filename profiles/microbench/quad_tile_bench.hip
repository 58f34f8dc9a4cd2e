// Micro-benchmark v2 of the LDS-tiled sparse x dense product with static register accumulators: "quad" layout.
//   hipcc --offload-arch=gfx950 -O3 -o quad_tile_bench quad_tile_bench.hip && ./quad_tile_bench
//
// slot_tile_bench.hip (v1) walks a per-slot loop over the slot's nonzeros of the tile: 7-10 ns per nonzero per CU, no
// better than the L2 row gather (9.3) — every nonzero pays scalar loop control, a taken branch and an exposed LDS latency.
// Here the per-slot structure is FIXED: every (slot, tile) pair owns QW record positions (zero-weight padding when it
// has fewer nonzeros; the rare extra ones go to an overflow list walked by a guarded per-slot loop afterwards). With
// QW = 4 and 16 slots per 64-lane set, the record of (slot q, position j) sits in lane 4 q + j of the set's registers:
// every v_readlane has an immediate lane, there is no scalar loop control and no branch in the main stream, and the
// reads of slot q+1 are issued before the FMAs of slot q (counted lgkmcnt waits).
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
typedef __attribute__((address_space(3))) void *lds_ptr_t;

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

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

// Layout per (group, tile):
//   qrow[((g * T + t) * NSET + b) * 64 + lane]  u8   panel row inside the tile
//   qval[  same index                        ]  u32  count (0 = padding)
//   desc[(g * T + t) * DW + ...]: dword 0 = first overflow record, dword 1 = number of overflow records,
//                                  dwords 2.. = 4-bit extra counts, 8 slots per dword
//   okey / oval: overflow records sorted by slot: (slot << 8 | row), count
constexpr int DW = 8;

// MODE bit 1: stage tile 0 only; bit 2: no log2
template <int S, int NW, int TR, int MODE, int PU>
__global__ __launch_bounds__(64 * NW) void quad_kernel(const uint32_t *__restrict__ desc, const uint8_t *__restrict__ qrow,
                                                       const uint32_t *__restrict__ qval, const uint16_t *__restrict__ okey,
                                                       const uint32_t *__restrict__ oval, uint32_t n_tiles,
                                                       const double *__restrict__ X, uint32_t ld, uint32_t l,
                                                       const double *__restrict__ sc_out, double *__restrict__ out) {
    constexpr int NSET = (S + 15) / 16;
    static_assert(S % 8 == 0 && S <= 48, "S");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t rowbytes = ld * 8u;
    const uint32_t tile_bytes = TR * rowbytes;
    const uint32_t n_chunks = tile_bytes / 1024u;

    auto stage = [&](uint32_t t, uint32_t buf) {
        const char *srcu = reinterpret_cast<const char *>(X + (size_t)t * TR * ld);
        char *dst = lds + buf * tile_bytes;
        const uint32_t l16 = lane * 16u;
        constexpr uint32_t CH = (TR + NW - 1) / NW; // chunks per wave at <= 1 KB rows
#pragma unroll
        for (uint32_t i = 0; i < CH; i++) {
            const uint32_t c = wave + i * NW;
            if (c < n_chunks) __builtin_amdgcn_global_load_lds(srcu + c * 1024u + l16, (lds_ptr_t)(dst + c * 1024u), 16, 0, 0);
        }
    };
    stage(0, 0);

    const uint32_t group = blockIdx.x * NW + wave;
    const uint32_t lcol16 = (lane * 2u < l ? lane : 0u) * 16u;
    d2 acc[S];
#pragma unroll
    for (int s = 0; s < S; s++) acc[s] = (d2){0.0, 0.0};
    // per-lane outer scale of each set: lane 4q+j belongs to slot 16b+q
    double so[NSET];
#pragma unroll
    for (int b = 0; b < NSET; b++) so[b] = sc_out[(size_t)group * S + min(16u * b + (lane >> 2), (uint32_t)(S - 1))];
    // overflow records gather their slot's scale from LDS
    double *wsc = reinterpret_cast<double *>(lds + 2u * tile_bytes) + wave * S;
    if (lane < S) wsc[lane] = sc_out[(size_t)group * S + lane];

    const size_t vbase = (size_t)group * n_tiles;
    uint32_t rr[NSET], rv[NSET];
#pragma unroll
    for (int b = 0; b < NSET; b++) {
        rr[b] = qrow[(vbase * NSET + b) * 64 + lane];
        rv[b] = qval[(vbase * NSET + b) * 64 + lane];
    }
    uint32_t vdesc = desc[vbase * DW + (lane & 7u)];
    uint32_t ok = 0, ov = 0;
    {
        const uint32_t o0 = rdlane(vdesc, 0);
        ok = okey[o0 + lane];
        ov = oval[o0 + lane];
    }
    uint32_t vdesc_nxt = desc[(vbase + min(1u, n_tiles - 1)) * DW + (lane & 7u)];

    for (uint32_t t = 0; t < n_tiles; t++) {
        __syncthreads(); // tile t is in buffer t&1
        const char *tile = lds + ((MODE & 2) ? 0u : (t & 1u) * tile_bytes) + lcol16;
        const bool more = t + 1 < n_tiles;
#pragma unroll
        for (int b = 0; b < NSET; b++) {
            const int nq = (S - 16 * b) < 16 ? (S - 16 * b) : 16; // slots in this set
            // lane-parallel decode + map
            const uint32_t v = rv[b];
            const double f = v == 0u ? 0.0 : ((MODE & 4) ? (double)v * so[b] : fast_log2(1.0 + (double)v * so[b]));
            const uint32_t voff = rr[b] * rowbytes;
            const uint32_t flo = (uint32_t)__double2loint(f), fhi = (uint32_t)__double2hiint(f);
            if (more) {
                rr[b] = qrow[((vbase + t + 1) * NSET + b) * 64 + lane];
                rv[b] = qval[((vbase + t + 1) * NSET + b) * 64 + lane];
            }
            if (b == NSET - 1 && !(MODE & 2) && more) stage(t + 1, (t + 1) & 1u);
            // software pipeline over units of PU records (PU = 2: half a slot, 4: a slot): the reads of unit u+1 are issued
            // before the FMAs of unit u
            constexpr int NU = 4 / PU;
            d2 x[2][PU];
            double w[2][PU];
#pragma unroll
            for (int u = 0; u <= nq * NU; u++) {
                if (u < nq * NU) {
#pragma unroll
                    for (int j = 0; j < PU; j++) {
                        const uint32_t off = (MODE & 32) ? (uint32_t)((PU * u + j) * 800) : rdlane(voff, PU * u + j);
                        w[u & 1][j] = (MODE & 16) ? 1.0 + (double)(PU * u + j) : __hiloint2double((int)rdlane(fhi, PU * u + j), (int)rdlane(flo, PU * u + j));
                        if (MODE & 64) {
                            x[u & 1][j] = (d2){(double)off, 1.0};
                        } else {
                            x[u & 1][j] = *reinterpret_cast<const d2 *>(tile + off);
                        }
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0); // reads of unit u are issued before the FMAs of unit u-1
                }
                if (u > 0) {
                    const int s = 16 * b + (u - 1) / NU;
#pragma unroll
                    for (int j = 0; j < PU; j++) {
                        if (MODE & 128) {
                            asm volatile("" ::"v"(x[(u - 1) & 1][j]), "s"(w[(u - 1) & 1][j]));
                        } else {
                            acc[s].x = fma(w[(u - 1) & 1][j], x[(u - 1) & 1][j].x, acc[s].x);
                            acc[s].y = fma(w[(u - 1) & 1][j], x[(u - 1) & 1][j].y, acc[s].y);
                        }
                    }
                    // pin the FMAs here (they are pure arithmetic: without a use the compiler sinks them to the end of
                    // the kernel and spills every row it has read meanwhile)
                    asm volatile("" : "+v"(acc[s].x), "+v"(acc[s].y));
                }
                asm volatile("" ::: "memory");     // keep the pipeline one unit deep: neither the IR passes nor the
                __builtin_amdgcn_sched_barrier(0); // machine scheduler may hoist later reads over this point
            }
        }
        // overflow: nonzeros beyond the 4th of a (slot, tile) pair, sorted by slot
        const uint32_t n_ov = rdlane(vdesc, 1);
        if (n_ov) {
            const uint32_t slot = min(ok >> 8, (uint32_t)(S - 1)), row = ok & 255u;
            const double sov = wsc[slot];
            const double f = (MODE & 4) ? (double)ov * sov : fast_log2(1.0 + (double)ov * sov);
            const uint32_t voff = row * rowbytes;
            const uint32_t flo = (uint32_t)__double2loint(f), fhi = (uint32_t)__double2hiint(f);
            uint32_t p = 0;
#pragma unroll
            for (int s = 0; s < S; s++) {
                uint32_t n = (rdlane(vdesc, 2 + (s >> 3)) >> ((s & 7) * 4)) & 15u;
                while (n) {
                    const uint32_t off = rdlane(voff, p);
                    const double wv = __hiloint2double((int)rdlane(fhi, p), (int)rdlane(flo, p));
                    const d2 xx = *reinterpret_cast<const d2 *>(tile + off);
                    acc[s].x = fma(wv, xx.x, acc[s].x);
                    acc[s].y = fma(wv, xx.y, acc[s].y);
                    p++;
                    n--;
                }
            }
        }
        if (more) {
            const uint32_t o0 = rdlane(vdesc_nxt, 0);
            ok = okey[o0 + lane];
            ov = oval[o0 + lane];
        }
        vdesc = vdesc_nxt;
        vdesc_nxt = desc[(vbase + min(t + 2, n_tiles - 1)) * DW + (lane & 7u)];
    }
    __syncthreads();
    if (lane * 2u < l) {
#pragma unroll
        for (int s = 0; s < S; s++) *reinterpret_cast<d2 *>(out + ((size_t)group * S + s) * ld + lane * 2u) = acc[s];
    }
}

// ------------------------------------------------------------------------------------------------------------------
struct Problem {
    uint32_t n_groups, n_tiles, S, NSET, TR, ld, l;
    std::vector<uint32_t> desc, qval, oval;
    std::vector<uint8_t> qrow;
    std::vector<uint16_t> okey;
    std::vector<double> X, sc;
    uint64_t nnz = 0, n_over = 0, clipped = 0;
};

static inline uint64_t xs(uint64_t &s) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
}

static void build(Problem &P, double density) {
    P.NSET = (P.S + 15) / 16;
    const size_t visits = (size_t)P.n_groups * P.n_tiles;
    P.desc.assign(visits * DW, 0u);
    P.qrow.assign(visits * P.NSET * 64, 0);
    P.qval.assign(visits * P.NSET * 64, 0u);
    P.okey.clear();
    P.oval.clear();
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    const double inv_log1m = 1.0 / log(1.0 - density);
    for (uint32_t g = 0; g < P.n_groups; g++) {
        for (uint32_t t = 0; t < P.n_tiles; t++) {
            const size_t vi = (size_t)g * P.n_tiles + t;
            uint32_t *d = &P.desc[vi * DW];
            d[0] = (uint32_t)P.okey.size();
            uint32_t n_ov = 0;
            for (uint32_t s = 0; s < P.S; s++) {
                uint32_t c = 0, extra = 0;
                for (double r = -1.0;;) {
                    const double u = ((double)(xs(rng) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
                    r += 1.0 + floor(log(u) * inv_log1m);
                    if (r >= (double)P.TR) break;
                    const uint32_t val = 1u + (uint32_t)(xs(rng) % 7u);
                    if (c < 4) {
                        const size_t qi = (vi * P.NSET + s / 16) * 64 + (s % 16) * 4 + c;
                        P.qrow[qi] = (uint8_t)r;
                        P.qval[qi] = val;
                    } else if (extra < 15 && n_ov < 64) {
                        P.okey.push_back((uint16_t)((s << 8) | (uint32_t)r));
                        P.oval.push_back(val);
                        extra++;
                        n_ov++;
                    } else {
                        P.clipped++;
                        continue;
                    }
                    c++;
                    P.nnz++;
                }
                d[2 + (s >> 3)] |= extra << ((s & 7) * 4);
            }
            d[1] = n_ov;
            P.n_over += n_ov;
        }
    }
    for (int i = 0; i < 128; i++) {
        P.okey.push_back(0);
        P.oval.push_back(0);
    }
    P.X.resize((size_t)P.n_tiles * P.TR * P.ld);
    for (auto &x : P.X) x = (double)(xs(rng) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    P.sc.resize((size_t)P.n_groups * P.S);
    for (auto &x : P.sc) x = 0.5 + (double)(xs(rng) >> 11) * (1.0 / 9007199254740992.0);
}

static void host_row(const Problem &P, uint32_t g, uint32_t s, bool use_log, std::vector<double> &r) {
    r.assign(P.l, 0.0);
    const double so = P.sc[(size_t)g * P.S + s];
    auto add = [&](uint32_t t, uint32_t row, uint32_t v) {
        const double f = use_log ? log2(1.0 + (double)v * so) : (double)v * so;
        const double *xr = &P.X[((size_t)t * P.TR + row) * P.ld];
        for (uint32_t c = 0; c < P.l; c++) r[c] = fma(f, xr[c], r[c]);
    };
    for (uint32_t t = 0; t < P.n_tiles; t++) {
        const size_t vi = (size_t)g * P.n_tiles + t;
        for (uint32_t j = 0; j < 4; j++) {
            const size_t qi = (vi * P.NSET + s / 16) * 64 + (s % 16) * 4 + j;
            if (P.qval[qi]) add(t, P.qrow[qi], P.qval[qi]);
        }
        const uint32_t *d = &P.desc[vi * DW];
        uint32_t pos = d[0];
        for (uint32_t q = 0; q < s; q++) pos += (d[2 + (q >> 3)] >> ((q & 7) * 4)) & 15u;
        const uint32_t n = (d[2 + (s >> 3)] >> ((s & 7) * 4)) & 15u;
        for (uint32_t j = 0; j < n; j++) add(t, P.okey[pos + j] & 255u, P.oval[pos + j]);
    }
}

struct Dev {
    uint32_t *desc, *qval, *oval;
    uint8_t *qrow;
    uint16_t *okey;
    double *X, *sc, *out;
};

template <int S, int NW, int TR, int MODE, int PU>
static void run(const char *name, const Problem &P, const Dev &D, int n_cu, bool check) {
    const size_t shmem = (size_t)2 * TR * P.ld * 8 + (size_t)NW * S * 8;
    auto kern = quad_kernel<S, NW, TR, MODE, PU>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const uint32_t n_wg = P.n_groups / NW;
    const dim3 grid(n_wg), block(64 * NW);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, grid, block, shmem, 0, D.desc, D.qrow, D.qval, D.okey, D.oval, P.n_tiles, D.X, P.ld, P.l, D.sc, D.out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 3;
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(kern, grid, block, shmem, 0, D.desc, D.qrow, D.qval, D.okey, D.oval, P.n_tiles, D.X, P.ld, P.l, D.sc, D.out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double nnz_used = (double)P.nnz * ((double)n_wg * NW / P.n_groups);
    printf("%-28s S=%2d NW=%2d TR=%3d  %8.3f ms  %6.2f ns/nnz/CU  (grid %u x %d, lds %zu KB)\n", name, S, NW, TR, ms,
           ms * 1e6 * n_cu / nnz_used, grid.x, 64 * NW, shmem >> 10);
    fflush(stdout);
    if (check && !(MODE & 2)) {
        std::vector<double> ho((size_t)P.n_groups * S * P.ld);
        CK(hipMemcpy(ho.data(), D.out, ho.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        std::vector<double> r;
        for (uint32_t i = 0; i < 48; i++) {
            const uint32_t g = (uint32_t)(((uint64_t)i * 2654435761u) % (n_wg * NW)), s = (i * 7u) % S;
            host_row(P, g, s, !(MODE & 4), r);
            for (uint32_t c = 0; c < P.l; c++) {
                const double dv = ho[((size_t)g * S + s) * P.ld + c];
                worst = std::max(worst, fabs(dv - r[c]) / (fabs(r[c]) + 1e-9));
            }
        }
        printf("   max rel deviation from the host evaluation %.3e\n", worst);
    }
}

template <int S, int NW, int TR>
static void suite(int n_cu, uint32_t l, uint32_t wg_rounds) {
    Problem P;
    P.S = S;
    P.TR = TR;
    P.l = l;
    P.ld = l;
    P.n_tiles = (33000 + TR - 1) / TR;
    P.n_groups = (uint32_t)n_cu * wg_rounds * NW;
    const double est = (double)P.n_groups * P.n_tiles * S * TR * 0.03;
    if (est > 2.0e8) P.n_tiles = std::max(8u, (uint32_t)(2.0e8 / ((double)P.n_groups * S * TR * 0.03)));
    build(P, 0.03);
    printf("S=%d NW=%d TR=%d: %u groups x %u tiles, nnz %llu (%.2f per slot-tile), overflow %.1f %%, clipped %llu\n", S, NW, TR, P.n_groups,
           P.n_tiles, (unsigned long long)P.nnz, (double)P.nnz / ((double)P.n_groups * P.n_tiles * S), 100.0 * P.n_over / P.nnz,
           (unsigned long long)P.clipped);
    Dev D;
    CK(hipMalloc(&D.desc, P.desc.size() * 4));
    CK(hipMalloc(&D.qrow, P.qrow.size()));
    CK(hipMalloc(&D.qval, P.qval.size() * 4));
    CK(hipMalloc(&D.okey, P.okey.size() * 2));
    CK(hipMalloc(&D.oval, P.oval.size() * 4));
    CK(hipMalloc(&D.X, P.X.size() * 8 + 4096));
    CK(hipMalloc(&D.sc, P.sc.size() * 8));
    CK(hipMalloc(&D.out, (size_t)P.n_groups * S * P.ld * 8));
    CK(hipMemcpy(D.desc, P.desc.data(), P.desc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.qrow, P.qrow.data(), P.qrow.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.qval, P.qval.data(), P.qval.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.okey, P.okey.data(), P.okey.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.oval, P.oval.data(), P.oval.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.X, P.X.data(), P.X.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.sc, P.sc.data(), P.sc.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(D.out, 0, (size_t)P.n_groups * S * P.ld * 8));
    run<S, NW, TR, 0, 2>("quad PU=2", P, D, n_cu, true);
    run<S, NW, TR, 6, 2>("tile0 nolog", P, D, n_cu, false);
    run<S, NW, TR, 6 + 16, 2>("tile0 nolog -wbcast", P, D, n_cu, false);
    run<S, NW, TR, 6 + 16 + 32, 2>("tile0 nolog -wbcast -offbcast", P, D, n_cu, false);
    run<S, NW, TR, 6 + 64, 2>("tile0 nolog -ldsread", P, D, n_cu, false);
    run<S, NW, TR, 6 + 128, 2>("tile0 nolog -fma", P, D, n_cu, false);
    run<S, NW, TR, 6 + 16 + 32 + 64, 2>("tile0 nolog fma only", P, D, n_cu, false);
    run<S, NW, TR, 6 + 16 + 32 + 128, 2>("tile0 nolog ldsread only", P, D, n_cu, false);
    CK(hipFree(D.desc));
    CK(hipFree(D.qrow));
    CK(hipFree(D.qval));
    CK(hipFree(D.okey));
    CK(hipFree(D.oval));
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
    suite<16, 16, 96>(n_cu, l, 2);
    suite<24, 12, 96>(n_cu, l, 2);
    suite<40, 8, 96>(n_cu, l, 2);
    return 0;
}
