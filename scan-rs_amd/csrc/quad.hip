// quad.hip — the LDS-staged form of the sparse x dense product (north star: "row-block coalesced triplet reads and
// LDS-staged dense panels"), the alternative to the L2-blocked row gather of kernels.hip for panels of up to 104 columns.
//
// The gather form pulls one 800-byte panel row per nonzero through the CU's texture addresser (64 B/clk, busy 88-96 %:
// DESIGN.md §4). Here the panel is walked in tiles of QT_TR rows that are staged in LDS once per workgroup (LDS-DMA, double
// buffered) and every row is read from LDS (ds_read_b128); what makes that possible without dynamically indexed registers
// (rounds 1 and 2, profiles/microbench/README.md) is a second layout of the matrix:
//   * a wave owns QT_S outer vectors ("slots") for the whole launch: 2 f64 accumulators = 4 VGPRs per lane per slot
//     (lane = column pair), destination registers of every FMA are compile-time constants because the code is unrolled
//     over the slots;
//   * every (slot, tile) pair owns exactly 4 record positions ("quad": panel row inside the tile u8 + count u32, zero count
//     = padding); the record of (slot q of a set of 16, position j) sits in lane 4 q + j of the set's registers, so the
//     lane-parallel decode (one map evaluation per lane) feeds the serial part through v_readlane with immediate lanes;
//   * the nonzeros beyond the 4th of a (slot, tile) pair (9 % at 3 % density and 96-row tiles) go to an overflow list per
//     (wave, tile), sorted by slot, walked in batches of 64 by a per-slot loop.
// MEASURED (1 M x 33 k x 3 %, l = 100, MI355X, tools/pass_bench.py): 65 ms per cell-major pass and 82 ms per gene-major pass
// against 40 / 43 ms of the L2-blocked gather — the quad stream alone runs at 22 / 26 ms (5.5 - 6.5 ns per nonzero per CU,
// 86 % of the nonzeros), but the 13 - 14 % of the nonzeros that overflow the quads cost 24 - 29 ms (a dependent per-record
// loop with one exposed LDS latency each), the tile staging 10 ms and the map evaluation of the padded lanes 4 - 12 ms.
// So this path is OPT-IN (scanrs_mat_set_spmm_path(m, 3)), kept as the worked-out form of the north star's "LDS-staged dense
// panels" with parity tests; the default stays the gather.
// Sums are bit-reproducible (fixed order: quad positions, then overflow, tiles ascending; parts of a split tile range
// are added in order). The layout costs 20 B per (outer vector, tile) + the overflow records, about one more copy of the
// matrix per orientation, and is built on the device on first use.
#include "common.hpp"
#include "device_map.hpp"

#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace scanrs {

namespace {

constexpr uint32_t QT_S = 40;    // slots per wave (16 + 16 + 8)
constexpr uint32_t QT_NSET = 3;  // sets of 16 slots
constexpr uint32_t QT_NW = 8;    // waves per workgroup (2 per SIMD, <= 256 VGPRs)
constexpr uint32_t QT_TR = 96;   // panel rows per tile
constexpr uint32_t QT_LMAX = 104; // 2 x 96 x 104 x 8 B = 159744 B of LDS for the two tile buffers

typedef __attribute__((address_space(3))) void *lds_ptr_t;

// ---- builder -------------------------------------------------------------------------------------------------------
// tb[o * (T + 1) + t] = offset inside outer vector o of its first nonzero with inner index >= t * QT_TR
__global__ void quad_bounds_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices, uint64_t n_outer,
                                   uint32_t n_tiles, uint32_t *__restrict__ tb) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_outer * (n_tiles + 1)) return;
    const uint64_t o = e / (n_tiles + 1);
    const uint32_t t = (uint32_t)(e % (n_tiles + 1));
    const uint64_t s = indptr[o], end = indptr[o + 1];
    const uint64_t key = (uint64_t)t * QT_TR;
    uint64_t lo = s, hi = end;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)indices[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    tb[e] = (uint32_t)(lo - s);
}
// overflow records of every (group, tile) visit
__global__ void quad_count_kernel(const uint32_t *__restrict__ tb, uint64_t n_outer, uint64_t n_groups, uint32_t n_tiles,
                                  unsigned long long *__restrict__ ov_count) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_groups * n_tiles) return;
    const uint64_t g = e / n_tiles;
    const uint32_t t = (uint32_t)(e % n_tiles);
    uint32_t n = 0;
    for (uint32_t sl = 0; sl < QT_S; sl++) {
        const uint64_t o = g * QT_S + sl;
        if (o >= n_outer) break;
        const uint32_t c = tb[o * (n_tiles + 1) + t + 1] - tb[o * (n_tiles + 1) + t];
        n += c > 4u ? c - 4u : 0u;
    }
    ov_count[e] = n;
}
__global__ void quad_fill_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                 const uint32_t *__restrict__ values, const uint32_t *__restrict__ tb, uint64_t n_outer,
                                 uint64_t n_groups, uint32_t n_tiles, const unsigned long long *__restrict__ ov_start,
                                 uint8_t *__restrict__ qrow, uint32_t *__restrict__ qval, unsigned long long *__restrict__ desc,
                                 uint16_t *__restrict__ okey, uint32_t *__restrict__ oval) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_groups * n_tiles) return;
    const uint64_t g = e / n_tiles;
    const uint32_t t = (uint32_t)(e % n_tiles);
    unsigned long long ov = ov_start[e];
    const unsigned long long ov0 = ov;
    for (uint32_t sl = 0; sl < QT_S; sl++) {
        const uint64_t o = g * QT_S + sl;
        const size_t q = ((e * QT_NSET + sl / 16u) * 64u) + (sl % 16u) * 4u;
        uint32_t c = 0;
        uint64_t base = 0;
        if (o < n_outer) {
            const uint32_t a = tb[o * (n_tiles + 1) + t];
            c = tb[o * (n_tiles + 1) + t + 1] - a;
            base = indptr[o] + a;
        }
        for (uint32_t j = 0; j < 4u; j++) {
            const bool have = j < c;
            qrow[q + j] = have ? (uint8_t)(indices[base + j] - t * QT_TR) : (uint8_t)0;
            qval[q + j] = have ? values[base + j] : 0u;
        }
        for (uint32_t j = 4u; j < c; j++) {
            okey[ov] = (uint16_t)((sl << 8) | (indices[base + j] - t * QT_TR));
            oval[ov] = values[base + j];
            ov++;
        }
    }
    desc[2 * e] = ov0;
    desc[2 * e + 1] = ov - ov0;
}

} // namespace

// the quad layout of one orientation
struct QuadLayout {
    uint64_t n_groups = 0, n_ov = 0;
    uint32_t n_tiles = 0;
    DevBuf<uint8_t> qrow;             // [group][tile][set][64]
    DevBuf<uint32_t> qval;            // same index
    DevBuf<unsigned long long> desc;  // [group][tile][2]: first overflow record, number of overflow records
    DevBuf<uint16_t> okey;            // (slot << 8 | row in tile), sorted by slot inside a (group, tile) visit; 128 entries of padding
    DevBuf<uint32_t> oval;
    double bytes() const { return (double)qrow.n + qval.n * 4.0 + desc.n * 8.0 + okey.n * 2.0 + oval.n * 4.0; }
};

QuadLayout *quad_layout_build(Storage &st, const SparseCopy &cp) {
    auto q = std::make_unique<QuadLayout>();
    q->n_groups = (cp.n_outer + QT_S - 1) / QT_S;
    q->n_tiles = (uint32_t)((cp.n_inner + QT_TR - 1) / QT_TR);
    const uint64_t visits = q->n_groups * q->n_tiles;
    if (visits == 0) return q.release();
    hipStream_t s = st.stream;
    DevBuf<uint32_t> tb(cp.n_outer * ((uint64_t)q->n_tiles + 1));
    hipLaunchKernelGGL(quad_bounds_kernel, dim3((unsigned)((tb.n + 255) / 256)), dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.n_outer,
                       q->n_tiles, tb.p);
    DevBuf<unsigned long long> ovc(visits + 1), ovs(visits + 1);
    SCANRS_HIP(hipMemsetAsync(ovc.p + visits, 0, 8, s));
    hipLaunchKernelGGL(quad_count_kernel, dim3((unsigned)((visits + 255) / 256)), dim3(256), 0, s, tb.p, cp.n_outer, q->n_groups, q->n_tiles,
                       ovc.p);
    size_t tmp_bytes = 0;
    SCANRS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, ovc.p, ovs.p, 0ull, (size_t)visits + 1, rocprim::plus<unsigned long long>(), s));
    DevBuf<char> tmp(std::max<size_t>(tmp_bytes, 16));
    SCANRS_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, ovc.p, ovs.p, 0ull, (size_t)visits + 1, rocprim::plus<unsigned long long>(), s));
    unsigned long long n_ov = 0;
    SCANRS_HIP(hipMemcpyAsync(&n_ov, ovs.p + visits, 8, hipMemcpyDeviceToHost, s));
    SCANRS_HIP(hipStreamSynchronize(s));
    q->n_ov = n_ov;
    q->qrow.alloc(visits * QT_NSET * 64);
    q->qval.alloc(visits * QT_NSET * 64);
    q->desc.alloc(visits * 2);
    q->okey.alloc(n_ov + 128);
    q->oval.alloc(n_ov + 128);
    SCANRS_HIP(hipMemsetAsync(q->okey.p + n_ov, 0, 128 * 2, s));
    SCANRS_HIP(hipMemsetAsync(q->oval.p + n_ov, 0, 128 * 4, s));
    hipLaunchKernelGGL(quad_fill_kernel, dim3((unsigned)((visits + 255) / 256)), dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.values.p, tb.p,
                       cp.n_outer, q->n_groups, q->n_tiles, ovs.p, q->qrow.p, q->qval.p, q->desc.p, q->okey.p, q->oval.p);
    SCANRS_HIP(hipGetLastError());
    SCANRS_HIP(hipStreamSynchronize(s)); // the temporaries are released on return
    if (trace_on())
        fprintf(stderr, "[scanrs trace] quad layout: %llu outer x %llu inner, %llu groups x %u tiles, nnz %llu, overflow records %llu (%.1f %%), %.2f GB\n",
                (unsigned long long)cp.n_outer, (unsigned long long)cp.n_inner, (unsigned long long)q->n_groups, q->n_tiles,
                (unsigned long long)cp.nnz, (unsigned long long)n_ov, 100.0 * (double)n_ov / (double)std::max<uint64_t>(1, cp.nnz), q->bytes() / 1e9);
    return q.release();
}
void quad_layout_free(QuadLayout *q) { delete q; }

// ---- product -------------------------------------------------------------------------------------------------------
namespace {

struct QuadArgs {
    const uint8_t *qrow;
    const uint32_t *qval;
    const unsigned long long *desc;
    const uint16_t *okey;
    const uint32_t *oval;
    uint64_t n_groups, n_outer, n_inner;
    uint32_t n_tiles;
};

// out_part[part][outer][:] = sum over the part's tiles of f(v, outer, inner) * X[inner, :]
__global__ __launch_bounds__(64 * QT_NW, 2) void spmm_quad_kernel(QuadArgs qa, DevMap map, const double *__restrict__ X, uint32_t ldx,
                                                                  uint32_t l, double *__restrict__ out_part, uint32_t ldo,
                                                                  uint64_t part_stride, uint32_t tiles_per_part) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t rowbytes = ldx * 8u;
    const uint32_t tile_bytes = QT_TR * rowbytes;
    const uint32_t t0 = blockIdx.y * tiles_per_part, t1 = min(qa.n_tiles, t0 + tiles_per_part);
    const uint64_t group_raw = (uint64_t)blockIdx.x * QT_NW + wave;
    const bool live = group_raw < qa.n_groups;
    const uint64_t group = live ? group_raw : qa.n_groups - 1; // idle waves shadow the last group (they still stage tiles and meet the barriers)
    const char *Xb = reinterpret_cast<const char *>(X);
    const uint64_t x_bytes = qa.n_inner * (uint64_t)rowbytes;

    // LDS-DMA staging of tile t into buffer `buf`: 1 KB per wave-instruction, the chunks dealt to the waves
    auto stage = [&](uint32_t t, uint32_t buf) {
        const uint64_t src0 = (uint64_t)t * tile_bytes;
        const uint64_t left = x_bytes - src0; // bytes of the panel from this tile on
        const uint32_t bytes = left < tile_bytes ? (uint32_t)left : tile_bytes;
        char *dst = lds + buf * tile_bytes;
        const uint32_t l16 = lane * 16u;
        constexpr uint32_t CH = (QT_TR * QT_LMAX * 8u / 1024u + QT_NW - 1) / QT_NW + 1;
#pragma unroll
        for (uint32_t i = 0; i < CH; i++) {
            const uint32_t c = wave + i * QT_NW;
            const uint32_t off = c * 1024u;
            if (off < bytes) { // wave-uniform
                if (off + l16 < bytes) // the last chunk of the last tile may be partial: lanes past the end stay out
                    __builtin_amdgcn_global_load_lds(Xb + src0 + off + l16, (lds_ptr_t)(dst + off), 16, 0, 0);
            }
        }
    };
    if (t0 < t1) stage(t0, 0);

    const uint32_t lcol16 = (lane * 2u < l ? lane : 0u) * 16u;
    d2 acc[QT_S];
#pragma unroll
    for (int sl = 0; sl < (int)QT_S; sl++) acc[sl] = (d2){0.0, 0.0};

    // a chain that starts with a ScaleAxis on the outer index: its factor is a per-lane constant of the set (lane 4 q + j <-> slot 16 b + q)
    const bool pre_outer = map.n > 0 && map.ops[0].kind == OP_SCALE_AXIS && map.ops[0].a_outer;
    const int map_start = pre_outer ? 1 : 0;
    double pre[QT_NSET];
    uint32_t outer_of[QT_NSET];
#pragma unroll
    for (uint32_t b = 0; b < QT_NSET; b++) {
        const uint64_t o = group * QT_S + 16u * b + (lane >> 2);
        outer_of[b] = (uint32_t)min(o, qa.n_outer - 1);
        pre[b] = pre_outer ? map.ops[0].a[outer_of[b]] : 1.0;
    }

    const size_t vbase = (size_t)group * qa.n_tiles;
    uint32_t rr[QT_NSET], rv[QT_NSET];
    unsigned long long dsc = 0, dsc_nxt = 0; // lanes 0 / 1: first overflow record, count
    uint32_t ok = 0, ov = 0;
    if (t0 < t1) {
#pragma unroll
        for (uint32_t b = 0; b < QT_NSET; b++) {
            rr[b] = qa.qrow[((vbase + t0) * QT_NSET + b) * 64u + lane];
            rv[b] = qa.qval[((vbase + t0) * QT_NSET + b) * 64u + lane];
        }
        dsc = qa.desc[(vbase + t0) * 2u + (lane & 1u)];
        dsc_nxt = qa.desc[(vbase + min(t0 + 1u, t1 - 1u)) * 2u + (lane & 1u)];
        const unsigned long long o0 = __shfl(dsc, 0, 64);
        ok = qa.okey[o0 + lane];
        ov = qa.oval[o0 + lane];
    }

    for (uint32_t t = t0; t < t1; t++) {
        __syncthreads(); // tile t is in buffer (t - t0) & 1 (the barrier's fence drains this wave's LDS-DMA); everyone is done with the other buffer
        const char *tile = lds + ((t - t0) & 1u) * tile_bytes + lcol16;
        const bool more = t + 1 < t1;
        const uint32_t inner0 = t * QT_TR;
#pragma unroll
        for (uint32_t b = 0; b < QT_NSET; b++) {
            // lane-parallel decode + map of the set's 64 record positions
            const uint32_t v = rv[b], row = rr[b];
            const uint64_t o = group * QT_S + 16u * b + (lane >> 2);
            double f = 0.0;
            if (v != 0u && o < qa.n_outer && live)
                f = eval_map_simple_from(map, map_start, pre_outer ? pre[b] * (double)v : (double)v, outer_of[b], inner0 + row);
            const uint32_t voff = row * rowbytes;
            const uint32_t flo = (uint32_t)__double2loint(f), fhi = (uint32_t)__double2hiint(f);
            if (more) { // refill the raw registers of this set with the next tile's records
                rr[b] = qa.qrow[((vbase + t + 1) * QT_NSET + b) * 64u + lane];
                rv[b] = qa.qval[((vbase + t + 1) * QT_NSET + b) * 64u + lane];
            }
            // the next tile's LDS-DMA goes out last (measured: issued right after the barrier the pass is 5 ms slower — every later
            // vmcnt wait of the visit then also waits for the 77 KB of the tile)
            if (b == QT_NSET - 1 && more) stage(t + 1, (t + 1 - t0) & 1u);
            // software pipeline over units of 2 records: the reads of unit u+1 are issued before the FMAs of unit u
            constexpr int NU_FULL = 32;
            const int nu = 2 * ((int)QT_S - 16 * (int)b < 16 ? (int)QT_S - 16 * (int)b : 16); // units of this set (compile-time after unrolling)
            d2 x[2][2];
            double w[2][2];
#pragma unroll
            for (int u = 0; u <= NU_FULL; u++) {
                if (u > nu) break;
                if (u < nu) {
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const uint32_t off = rdlane(voff, 2 * u + j);
                        w[u & 1][j] = __hiloint2double((int)rdlane(fhi, 2 * u + j), (int)rdlane(flo, 2 * u + j));
                        x[u & 1][j] = *reinterpret_cast<const d2 *>(tile + off);
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0); // reads of unit u are issued before the FMAs of unit u-1
                }
                if (u > 0) {
                    const int sl = 16 * (int)b + (u - 1) / 2;
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        acc[sl].x = fma(w[(u - 1) & 1][j], x[(u - 1) & 1][j].x, acc[sl].x);
                        acc[sl].y = fma(w[(u - 1) & 1][j], x[(u - 1) & 1][j].y, acc[sl].y);
                    }
                    // pin the FMAs here (pure arithmetic: without a use the compiler sinks them to the end of the kernel and spills
                    // every row it has read meanwhile)
                    asm volatile("" : "+v"(acc[sl].x), "+v"(acc[sl].y));
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // overflow: the nonzeros beyond the 4th of a (slot, tile) pair, sorted by slot, in batches of 64
        const uint32_t n_ov = (uint32_t)__shfl(dsc, 1, 64);
        const unsigned long long ov_first = __shfl(dsc, 0, 64);
        for (uint32_t ob = 0; ob < n_ov; ob += 64u) {
            if (ob > 0) {
                ok = qa.okey[ov_first + ob + lane];
                ov = qa.oval[ov_first + ob + lane];
            }
            const uint32_t nb = min(64u, n_ov - ob);
            const uint32_t slot = min((uint32_t)(ok >> 8), QT_S - 1u), row = ok & 255u;
            const uint64_t o = group * QT_S + slot;
            double f = 0.0;
            if (lane < nb && o < qa.n_outer && live) f = eval_map_simple_from(map, 0, (double)ov, (uint32_t)o, inner0 + row);
            const uint32_t voff = row * rowbytes;
            const uint32_t flo = (uint32_t)__double2loint(f), fhi = (uint32_t)__double2hiint(f);
            const uint32_t vslot = lane < nb ? slot : 0xFFFFu;
            uint32_t p = 0;
#pragma unroll
            for (int sl = 0; sl < (int)QT_S; sl++) {
                while (p < nb && rdlane(vslot, p) == (uint32_t)sl) {
                    const uint32_t off = rdlane(voff, p);
                    const double wv = __hiloint2double((int)rdlane(fhi, p), (int)rdlane(flo, p));
                    const d2 xx = *reinterpret_cast<const d2 *>(tile + off);
                    acc[sl].x = fma(wv, xx.x, acc[sl].x);
                    acc[sl].y = fma(wv, xx.y, acc[sl].y);
                    p++;
                }
            }
        }
        if (more) { // the next tile's first overflow batch
            const unsigned long long o0 = __shfl(dsc_nxt, 0, 64);
            ok = qa.okey[o0 + lane];
            ov = qa.oval[o0 + lane];
        }
        dsc = dsc_nxt;
        dsc_nxt = qa.desc[(vbase + min(t + 2u, t1 - 1u)) * 2u + (lane & 1u)];
    }
    if (live && lane * 2u < l) {
        double *dst = out_part + (size_t)blockIdx.y * part_stride;
#pragma unroll
        for (int sl = 0; sl < (int)QT_S; sl++) {
            const uint64_t o = group * QT_S + sl;
            if (o < qa.n_outer) *reinterpret_cast<d2 *>(dst + o * ldo + lane * 2u) = acc[sl];
        }
    }
}

// out[o, :] = sum over parts (in order) + LowRankOffset term  (sqz/src/low_rank_offset.rs:76-80)
__global__ void quad_finish_kernel(const double *__restrict__ parts, uint32_t n_parts, uint64_t part_stride, uint64_t n_outer, uint32_t l,
                                   uint32_t ldo, double *__restrict__ out, const double *__restrict__ off_a, uint32_t rank,
                                   const double *__restrict__ off_w, uint32_t ldw) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_outer * l) return;
    const uint64_t o = e / l;
    const uint32_t c = (uint32_t)(e % l);
    double s = parts[o * ldo + c];
    for (uint32_t p = 1; p < n_parts; p++) s += parts[(size_t)p * part_stride + o * ldo + c];
    for (uint32_t q = 0; q < rank; q++) s += off_a[o * rank + q] * off_w[(size_t)q * ldw + c];
    out[o * ldo + c] = s;
}

} // namespace

bool spmm_quad_ok(const SparseCopy &cp, const DevMap &map, uint32_t ldx, uint32_t l) {
    return map_is_simple(map) && l >= 16 && l <= QT_LMAX && ldx <= QT_LMAX && (ldx & 1u) == 0 && cp.n_outer > 0 && cp.n_inner > 0 && cp.nnz > 0;
}

void launch_spmm_quad(Storage &st, SparseCopy &cp, QuadLayout &q, const DevMap &map, const double *X, uint32_t ldx, uint32_t l,
                      double *out, uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w, uint32_t ldw) {
    if ((ldx & 1u) || (ldo & 1u)) fail(SCANRS_ERR_ARGUMENT, "panel leading dimensions must be even");
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t wgs = (uint32_t)((q.n_groups + QT_NW - 1) / QT_NW);
    // enough workgroups for ~4 per CU: a short outer dimension (genes) splits the tile range into parts whose sums are added in order
    uint32_t parts = 1;
    if (wgs < 4u * (uint32_t)n_cu) parts = std::min<uint32_t>(q.n_tiles, (4u * (uint32_t)n_cu + wgs - 1) / wgs);
    const uint32_t tpp = (q.n_tiles + parts - 1) / parts;
    parts = (q.n_tiles + tpp - 1) / tpp;
    const uint64_t part_stride = cp.n_outer * (uint64_t)ldo;
    const bool direct = parts == 1 && rank == 0;
    double *dst = direct ? out : st.scratch.get<double>("quad_parts", (size_t)parts * part_stride);
    QuadArgs qa{q.qrow.p, q.qval.p, q.desc.p, q.okey.p, q.oval.p, q.n_groups, cp.n_outer, cp.n_inner, q.n_tiles};
    const size_t shmem = (size_t)2 * QT_TR * ldx * 8;
    SCANRS_HIP(hipFuncSetAttribute((const void *)spmm_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const double bytes = q.bytes() + (double)cp.n_inner * l * 8.0 + (double)cp.n_outer * l * 8.0;
    if (st.prof.on)
        st.prof.begin(st.stream, cp.n_outer >= cp.n_inner ? "spmm_quad_kernel/long-outer" : "spmm_quad_kernel/short-outer", bytes,
                      ((double)q.n_groups * q.n_tiles * QT_S * 4.0 + (double)q.n_ov) * 8.0 * l);
    hipLaunchKernelGGL(spmm_quad_kernel, dim3(wgs, parts), dim3(64 * QT_NW), shmem, st.stream, qa, map, X, ldx, l, dst, ldo, part_stride, tpp);
    if (st.prof.on) st.prof.end(st.stream);
    if (!direct) {
        const uint64_t n = cp.n_outer * (uint64_t)l;
        hipLaunchKernelGGL(quad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st.stream, dst, parts, part_stride, cp.n_outer, l,
                           ldo, out, off_a, rank, off_w, ldw);
    }
    SCANRS_HIP(hipGetLastError());
}

} // namespace scanrs
