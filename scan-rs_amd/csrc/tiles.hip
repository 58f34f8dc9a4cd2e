// tiles.hip — the hybrid sparse x dense product for panels of up to 104 columns: LDS-staged panel tiles for the bulk of the
// nonzeros AND the L2-blocked row gather (kernels.hip) for the rest, both at the same time.
//
// Why two pipes. The gather form pulls one 800-byte panel row per nonzero through the CU's texture addresser (64 B/clk,
// busy 91-96 % of a launch: DESIGN.md section 4) while the LDS (256 B/clk for ds_read_b128) idles; an LDS-only form (round 2,
// quad.hip) pays for every nonzero that does not fit its fixed record structure. Here the matrix is SPLIT once per map:
//   * tile part: every (outer vector, tile of TL_T = 96 panel rows) pair owns exactly K record positions (K = 2..4); its
//     first K nonzeros go there as (row inside the tile u8, weight f64) — the weight is the whole map chain evaluated
//     once at build time (materialized, so a padded position costs no logarithm) — and unused positions hold weight 0;
//   * overflow part: the nonzeros beyond the K-th of a pair form an ordinary compressed matrix of the same shape (indptr /
//     indices / f64 weights) that the existing L2-blocked gather kernel walks unchanged.
// product = tile kernel (this file) + gather over the overflow part, launched on two streams so that their workgroups share
// the CUs: the tile kernel is PERSISTENT (one workgroup of 8 waves per CU, resident for the whole product, 2 x 77 KB of LDS,
// <= 184 VGPRs so that 2 waves / SIMD leave registers for gather waves); the gather kernel's workgroups fill what is left of
// every SIMD. The tile kernel keeps the LDS pipe and the f64 FMA pipe busy, the gather kernel the texture addresser.
//
// Tile kernel: a wave owns S outer vectors ("slots") of one group for a whole item (a range of tiles): 2 f64 accumulators
// per lane per slot (lane = column pair), every FMA destination a compile-time register because the code is unrolled over
// the positions; the 64 lanes of a record set hold the set's positions (slot q, position j -> lane q K + j), so the serial
// part reads them with v_readlane at immediate lanes; panel tiles are staged by LDS-DMA (global_load_lds, double buffered,
// one barrier per tile); ds_read_b128 of position p + W is issued before the FMAs of position p (W reads in flight).
// Sums are bit-reproducible: positions in order, tiles ascending, parts and the overflow sum added in a fixed order.
#include "common.hpp"
#include "device_map.hpp"

#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace scanrs {

namespace {

constexpr uint32_t TL_T = 96;     // panel rows per tile
constexpr uint32_t TL_NW = 8;     // waves per workgroup (2 per SIMD)
constexpr uint32_t TL_LMAX = 104; // 2 x 96 x 104 x 8 B = 159744 B of LDS for the two tile buffers
constexpr int TL_W = 6;           // LDS row reads in flight per wave

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef __attribute__((address_space(3))) const char *lds_cptr_t;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// ---- builder -------------------------------------------------------------------------------------------------------
// One workgroup per outer vector. tb[o * (nt + 1) + t] = offset inside the vector of its first nonzero with inner index
// >= t * TL_T; ovtot[o] = number of its nonzeros beyond the K-th of a tile.
__global__ __launch_bounds__(256) void tile_bounds_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                          uint64_t n_outer, uint32_t nt, uint32_t K, uint32_t *__restrict__ tb,
                                                          unsigned long long *__restrict__ ovtot) {
    const uint64_t o = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (o >= n_outer) return;
    const uint32_t tid = threadIdx.x;
    const uint64_t s = indptr[o];
    const uint32_t len = (uint32_t)(indptr[o + 1] - s);
    uint32_t *row = tb + o * ((uint64_t)nt + 1);
    for (uint32_t p = tid; p < len; p += 256u) {
        const uint32_t tc = indices[s + p] / TL_T;
        const uint32_t tp = p ? indices[s + p - 1] / TL_T + 1u : 0u;
        for (uint32_t t = tp; t <= tc; t++) row[t] = p;
    }
    const uint32_t tl = len ? indices[s + len - 1] / TL_T + 1u : 0u;
    for (uint32_t t = tl + tid; t <= nt; t += 256u) row[t] = len;
    __syncthreads();
    uint32_t ov = 0;
    for (uint32_t t = tid; t < nt; t += 256u) {
        const uint32_t c = row[t + 1] - row[t];
        ov += c > K ? c - K : 0u;
    }
    __shared__ uint32_t ws[4];
    const uint32_t lane = tid & 63u;
    ov = wave_incl_scan(ov, lane);
    if (lane == 63u) ws[tid >> 6] = ov;
    __syncthreads();
    if (tid == 0) ovtot[o] = (unsigned long long)ws[0] + ws[1] + ws[2] + ws[3];
}

// One workgroup per outer vector: fills the vector's record positions of every tile and its overflow entries.
// Record index of (group g, tile t, set b, lane): ((g nt + t) nset + b) 64 + lane, lane = q K + j for slot 16.. q of the set.
__global__ __launch_bounds__(256) void tile_fill_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                        const uint32_t *__restrict__ values, const uint32_t *__restrict__ tb, uint64_t n_outer,
                                                        uint32_t nt, uint32_t K, uint32_t S, uint32_t sps, uint32_t nset, DevMap map,
                                                        const unsigned long long *__restrict__ ov_indptr, uint8_t *__restrict__ prow,
                                                        double *__restrict__ pw, uint32_t *__restrict__ ov_indices, double *__restrict__ ov_w) {
    const uint64_t o = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (o >= n_outer) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t s = indptr[o];
    const uint32_t *row = tb + o * ((uint64_t)nt + 1);
    const uint64_t g = o / S;
    const uint32_t sl = (uint32_t)(o % S), b = sl / sps, q = sl % sps;
    const unsigned long long ov0 = ov_indptr[o];
    __shared__ uint32_t ws[4];
    uint32_t carry = 0; // overflow entries of the tiles before this chunk (same value in every thread)
    for (uint32_t t0 = 0; t0 < nt; t0 += 256u) {
        const uint32_t t = t0 + tid;
        uint32_t a = 0, c = 0;
        if (t < nt) {
            a = row[t];
            c = row[t + 1] - a;
        }
        const uint32_t ovc = c > K ? c - K : 0u;
        const uint32_t incl = wave_incl_scan(ovc, lane);
        __syncthreads(); // ws of the previous chunk has been read
        if (lane == 63u) ws[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wave; w++) before += ws[w];
        before += incl - ovc;
        carry += ws[0] + ws[1] + ws[2] + ws[3];
        if (t < nt) {
            const uint64_t base = ((g * nt + t) * nset + b) * 64u + (uint64_t)q * K;
            const uint32_t kept = c < K ? c : K;
            for (uint32_t j = 0; j < kept; j++) {
                const uint32_t idx = indices[s + a + j];
                prow[base + j] = (uint8_t)(idx - t * TL_T);
                pw[base + j] = eval_map(map, values[s + a + j], (uint32_t)o, idx);
            }
            unsigned long long op = ov0 + before;
            for (uint32_t j = K; j < c; j++, op++) {
                const uint32_t idx = indices[s + a + j];
                ov_indices[op] = idx;
                ov_w[op] = eval_map(map, values[s + a + j], (uint32_t)o, idx);
            }
        }
    }
}

} // namespace

// the tile layout of one orientation under one map
struct TileLayout {
    uint32_t K = 0, S = 0, sps = 0, nset = 0, n_tiles = 0;
    uint64_t n_groups = 0;
    DevBuf<uint8_t> prow; // [group][tile][set][64]
    DevBuf<double> pw;    // same index
    SparseCopy ov;        // the overflow part: indptr / indices / fvals (weights); `values` stays empty
    // identity of the map the weights were evaluated under (MapOp ids are never reused)
    int sig_n = -1;
    uint32_t sig_id[MAX_OPS] = {};
    int sig_outer[MAX_OPS] = {};
    double bytes() const { return (double)prow.n + (double)pw.n * 8.0 + (double)ov.nnz * 12.0; }
    bool matches(const DevMap &map, uint32_t k, uint32_t s) const {
        if (sig_n != map.n || K != k || S != s) return false;
        for (int i = 0; i < map.n; i++)
            if (sig_id[i] != map.ops[i].id || sig_outer[i] != map.ops[i].a_outer) return false;
        return true;
    }
};

void tile_layout_free(TileLayout *t) { delete t; }

uint32_t ensure_bounds_public(Storage &st, SparseCopy &cp); // kernels.hip

TileLayout *tile_layout_build(Storage &st, const SparseCopy &cp, const DevMap &map, uint32_t K, uint32_t S) {
    Tick tick("tile layout build");
    auto tl = std::make_unique<TileLayout>();
    tl->K = K;
    tl->S = S;
    tl->sps = 64u / K;
    tl->nset = (S + tl->sps - 1) / tl->sps;
    tl->n_groups = (cp.n_outer + S - 1) / S;
    tl->n_tiles = (uint32_t)((cp.n_inner + TL_T - 1) / TL_T);
    tl->sig_n = map.n;
    for (int i = 0; i < map.n; i++) {
        tl->sig_id[i] = map.ops[i].id;
        tl->sig_outer[i] = map.ops[i].a_outer;
    }
    hipStream_t s = st.stream;
    const uint32_t nt = tl->n_tiles;
    const uint64_t n_rec = tl->n_groups * nt * tl->nset * 64u;
    if (n_rec == 0) return tl.release();
    DevBuf<uint32_t> tb(cp.n_outer * ((uint64_t)nt + 1));
    DevBuf<unsigned long long> ovtot(cp.n_outer + 1), ovptr(cp.n_outer + 1);
    SCANRS_HIP(hipMemsetAsync(ovtot.p + cp.n_outer, 0, 8, s));
    const dim3 grid((unsigned)std::min<uint64_t>(cp.n_outer, 1u << 20), (unsigned)((cp.n_outer + (1u << 20) - 1) >> 20));
    hipLaunchKernelGGL(tile_bounds_kernel, grid, dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.n_outer, nt, K, tb.p, ovtot.p);
    size_t tmp_bytes = 0;
    SCANRS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, ovtot.p, ovptr.p, 0ull, (size_t)cp.n_outer + 1, rocprim::plus<unsigned long long>(), s));
    DevBuf<char> tmp(std::max<size_t>(tmp_bytes, 16));
    SCANRS_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, ovtot.p, ovptr.p, 0ull, (size_t)cp.n_outer + 1, rocprim::plus<unsigned long long>(), s));
    unsigned long long n_ov = 0;
    SCANRS_HIP(hipMemcpyAsync(&n_ov, ovptr.p + cp.n_outer, 8, hipMemcpyDeviceToHost, s));
    SCANRS_HIP(hipStreamSynchronize(s));
    tl->prow.alloc(n_rec);
    tl->pw.alloc(n_rec);
    SCANRS_HIP(hipMemsetAsync(tl->prow.p, 0, n_rec, s));
    SCANRS_HIP(hipMemsetAsync(tl->pw.p, 0, n_rec * 8, s));
    SparseCopy &ov = tl->ov;
    ov.n_outer = cp.n_outer;
    ov.n_inner = cp.n_inner;
    ov.nnz = n_ov;
    ov.indptr.alloc(cp.n_outer + 1);
    SCANRS_HIP(hipMemcpyAsync(ov.indptr.p, ovptr.p, (cp.n_outer + 1) * 8, hipMemcpyDeviceToDevice, s));
    ov.indices.alloc(std::max<uint64_t>(n_ov, 1));
    ov.fvals.alloc(std::max<uint64_t>(n_ov, 1));
    hipLaunchKernelGGL(tile_fill_kernel, grid, dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.values.p, tb.p, cp.n_outer, nt, K, S, tl->sps,
                       tl->nset, map, ovptr.p, tl->prow.p, tl->pw.p, ov.indices.p, ov.fvals.p);
    SCANRS_HIP(hipGetLastError());
    if (n_ov) ensure_bounds_public(st, ov);
    SCANRS_HIP(hipStreamSynchronize(s)); // the temporaries are released on return
    if (trace_on())
        fprintf(stderr, "[scanrs trace] tile layout: %llu outer x %llu inner, K %u, S %u, %llu groups x %u tiles, nnz %llu, overflow %llu (%.1f %%), slot use %.1f %%, %.2f GB\n",
                (unsigned long long)cp.n_outer, (unsigned long long)cp.n_inner, K, S, (unsigned long long)tl->n_groups, nt,
                (unsigned long long)cp.nnz, (unsigned long long)n_ov, 100.0 * (double)n_ov / (double)std::max<uint64_t>(1, cp.nnz),
                100.0 * (double)(cp.nnz - n_ov) / ((double)cp.n_outer * nt * K), tl->bytes() / 1e9);
    return tl.release();
}

// ---- product -------------------------------------------------------------------------------------------------------
namespace {

struct TileArgs {
    const uint8_t *prow;
    const double *pw;
    uint64_t n_groups, n_outer, n_inner;
    uint32_t n_tiles;
};

// parts[part][outer][:] = sum over the part's tiles of weight * X[inner, :]
// The VGPR cap is what lets gather waves share the SIMDs: 2 tile waves x 184 leave 144 registers per SIMD (S = 32), 2 x 168
// leave 176 (S = 28).
template <int K, int S>
__device__ __forceinline__ void spmm_tile_body(const TileArgs &ta, const double *__restrict__ X, uint32_t ldx, uint32_t l,
                                               double *__restrict__ parts, uint32_t ldo, uint64_t part_stride, uint32_t n_parts,
                                               uint32_t tiles_per_part, uint32_t n_items) {
    constexpr int SPS = 64 / K;
    constexpr int NSET = (S + SPS - 1) / SPS;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t rowbytes = ldx * 8u;
    const uint32_t tile_bytes = TL_T * rowbytes;
    const char *Xb = reinterpret_cast<const char *>(X);
    const uint64_t x_bytes = ta.n_inner * (uint64_t)rowbytes;

    // LDS-DMA staging of tile t into buffer `buf`: 1 KB per wave-instruction, chunk i of this wave. No branches (a join
    // makes the compiler wait for ALL outstanding LDS reads at the next use): every wave issues CH chunks per tile, chunk
    // numbers past the tile's last one repeat that one (same bytes to the same place), lanes past the end of the panel or of
    // the tile re-read the last 16 valid bytes — the buffers are whole KBs apart, so what they write is never read.
    constexpr int CH = (int)((TL_T * TL_LMAX * 8u / 1024u + TL_NW - 1) / TL_NW); // chunks per wave and tile
    const uint32_t n_chunks = (tile_bytes + 1023u) / 1024u;
    const uint32_t buf_stride = n_chunks * 1024u;
    auto stage_chunk = [&](uint32_t t, uint32_t buf, uint32_t i) {
        const uint32_t off = min(wave + i * TL_NW, n_chunks - 1u) * 1024u;
        const uint64_t src0 = (uint64_t)t * tile_bytes;
        const uint64_t left = x_bytes - src0; // bytes of the panel from this tile on
        const uint32_t last16 = (left < tile_bytes ? (uint32_t)left : tile_bytes) - 16u;
        char *dst = lds + buf * buf_stride;
        uint32_t l16 = lane * 16u;
        asm volatile("" : "+v"(l16)); // recomputed per chunk: the per-chunk offsets must not stay in registers across the loop
        // Hand-issued (M0 = LDS destination of lane 0, 16 bytes per lane): with the builtin anywhere in the block of the
        // position pipeline the compiler stops counting LDS reads and drains them all (lgkmcnt(0)) at every use.
        const uint32_t m0v = (uint32_t)(uintptr_t)(lds_ptr_t)(dst + off);
        const uint32_t voff32 = min(off + l16, last16);
        const char *sbase = Xb + src0;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff32), "s"(sbase) : "memory");
    };
    const uint32_t lcol16 = (lane * 2u < l ? lane : 0u) * 16u;

    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint32_t part = item % n_parts, wgg = item / n_parts;
        const uint32_t t0 = part * tiles_per_part, t1 = min(ta.n_tiles, t0 + tiles_per_part);
        const uint64_t group_raw = (uint64_t)wgg * TL_NW + wave;
        const bool live = group_raw < ta.n_groups;
        const uint64_t group = live ? group_raw : ta.n_groups - 1; // idle waves shadow the last group (they still stage tiles and meet the barriers)
        __syncthreads(); // everyone is done with the buffers of the previous item
        if (t0 < t1) {
#pragma unroll
            for (int i = 0; i < CH; i++) stage_chunk(t0, 0, i);
        }

        d2 acc[S];
#pragma unroll
        for (int sl = 0; sl < S; sl++) acc[sl] = (d2){0.0, 0.0};

        const size_t vbase = (size_t)group * ta.n_tiles;
        // record rows are bytes; a lane loads the aligned dword that holds its byte and extracts it when the set is worked
        // (a byte load is zero-extended right behind the load, and that wait would also cover the tile's LDS-DMA)
        const uint32_t *prow32 = reinterpret_cast<const uint32_t *>(ta.prow);
        const uint32_t rsh = (lane & 3u) * 8u;
        uint32_t crow[NSET], nrow[NSET];
        double cw[NSET], nw[NSET];
#pragma unroll
        for (int b = 0; b < NSET; b++) {
            crow[b] = 0;
            cw[b] = 0.0;
            if (t0 < t1) {
                crow[b] = prow32[((vbase + t0) * NSET + b) * 16u + (lane >> 2)];
                cw[b] = ta.pw[((vbase + t0) * NSET + b) * 64u + lane];
            }
        }

        for (uint32_t t = t0; t < t1; t++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's LDS-DMA chunks of tile t (the compiler does not see them)
            __syncthreads(); // tile t is in buffer (t - t0) & 1; everyone is done with the other buffer
            const lds_cptr_t tile = (lds_cptr_t)(lds + ((t - t0) & 1u) * buf_stride + lcol16);
            const uint32_t tn = t + 1 < t1 ? t + 1 : t; // the last visit of an item re-loads itself: no branch in the loop
            // the next tile's records: nothing in this visit waits for them
#pragma unroll
            for (int b = 0; b < NSET; b++) {
                nrow[b] = prow32[((vbase + tn) * NSET + b) * 16u + (lane >> 2)];
                nw[b] = ta.pw[((vbase + tn) * NSET + b) * 64u + lane];
            }
            // The visit's S K positions as one software pipeline over position g (set g / (SPS K), lane g % (SPS K)): per step
            //   R(g)          three v_readlane: row offset and the weight's halves into SGPRs
            //   A(g - 1)      LDS address of the row
            //   L(g - 2)      ds_read_b128 of the row
            //   F(g - 2 - W)  the two FMAs
            // so that no instruction of a step depends on another one of the same step (a wave issues in order and two waves
            // share a SIMD: dependent neighbours leave the vector unit idle), and the next tile's LDS-DMA chunks are issued a
            // few at a time between the steps instead of as one burst that queues on the texture path.
            uint32_t voff[NSET], wlo[NSET], whi[NSET];
#pragma unroll
            for (int b = 0; b < NSET; b++) {
                voff[b] = ((crow[b] >> rsh) & 255u) * rowbytes;
                wlo[b] = (uint32_t)__double2loint(cw[b]);
                whi[b] = (uint32_t)__double2hiint(cw[b]);
            }
            constexpr int PPS = SPS * K;      // positions per full set
            constexpr int NPT = S * K;        // positions per visit
            constexpr int WR = TL_W + 3;      // weights live from R(g) to F(g)
            constexpr int DM = NPT / CH;      // steps between two LDS-DMA chunks
            uint32_t offs[2];
            lds_cptr_t addr[2];
            double wq[WR];
            d2 x[TL_W];
#pragma unroll
            for (int i = 0; i < NPT + TL_W + 2; i++) {
                if (i >= 2 + TL_W) { // F(i - 2 - W)
                    const int g = i - 2 - TL_W;
                    const int sl = (g / PPS) * SPS + (g % PPS) / K;
                    acc[sl].x = fma(wq[g % WR], x[g % TL_W].x, acc[sl].x);
                    acc[sl].y = fma(wq[g % WR], x[g % TL_W].y, acc[sl].y);
                    // pin the FMAs here (pure arithmetic: without a use the compiler sinks them to the end of the kernel)
                    asm volatile("" : "+v"(acc[sl].x), "+v"(acc[sl].y));
                }
                if (i >= 2 && i - 2 < NPT) { // L(i - 2)
                    const int g = i - 2;
                    x[g % TL_W] = *(const __attribute__((address_space(3))) d2 *)addr[g % 2];
                }
                if (i >= 1 && i - 1 < NPT) { // A(i - 1)
                    const int g = i - 1;
                    addr[g % 2] = tile + offs[g % 2];
                    asm volatile("" : "+v"(addr[g % 2])); // the address now, not in front of the read
                }
                if (i < NPT) { // R(i)
                    const int b = i / PPS, p = i % PPS;
                    offs[i % 2] = rdlane(voff[b], p);
                    wq[i % WR] = __hiloint2double((int)rdlane(whi[b], p), (int)rdlane(wlo[b], p));
                }
                if (i % DM == DM / 2 && i / DM < CH) stage_chunk(tn, (t + 1 - t0) & 1u, i / DM);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int b = 0; b < NSET; b++) {
                crow[b] = nrow[b];
                cw[b] = nw[b];
            }
        }
        if (live && lane * 2u < l) {
            double *dst = parts + (size_t)part * part_stride;
#pragma unroll
            for (int sl = 0; sl < S; sl++) {
                const uint64_t o = group * S + sl;
                if (o < ta.n_outer) *reinterpret_cast<d2 *>(dst + o * ldo + lane * 2u) = acc[sl];
            }
        }
    }
}

template <int K, int S>
__global__ __launch_bounds__(64 * TL_NW, 2) void spmm_tile_kernel(
    TileArgs ta, const double *__restrict__ X, uint32_t ldx, uint32_t l, double *__restrict__ parts, uint32_t ldo, uint64_t part_stride,
    uint32_t n_parts, uint32_t tiles_per_part, uint32_t n_items) {
    spmm_tile_body<K, S>(ta, X, ldx, l, parts, ldo, part_stride, n_parts, tiles_per_part, n_items);
}
template <int K, int S>
__global__ __launch_bounds__(64 * TL_NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void spmm_tile_kernel_r168(
    TileArgs ta, const double *__restrict__ X, uint32_t ldx, uint32_t l, double *__restrict__ parts, uint32_t ldo, uint64_t part_stride,
    uint32_t n_parts, uint32_t tiles_per_part, uint32_t n_items) {
    spmm_tile_body<K, S>(ta, X, ldx, l, parts, ldo, part_stride, n_parts, tiles_per_part, n_items);
}

// out[o, :] = sum over parts (in order) + overflow sum + LowRankOffset term  (sqz/src/low_rank_offset.rs:76-80)
__global__ __launch_bounds__(256) void tile_finish_kernel(const double *__restrict__ parts, uint32_t n_parts, uint64_t part_stride,
                                                          const double *__restrict__ ovout, uint64_t n_outer, uint32_t l, uint32_t ldo,
                                                          double *__restrict__ out, const double *__restrict__ off_a, uint32_t rank,
                                                          const double *__restrict__ off_w, uint32_t ldw) {
    const uint32_t hp = (l + 1u) / 2u; // column pairs
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_outer * hp) return;
    const uint64_t o = e / hp;
    const uint32_t c = (uint32_t)(e % hp) * 2u;
    d2 s = *reinterpret_cast<const d2 *>(parts + o * ldo + c);
    for (uint32_t p = 1; p < n_parts; p++) {
        const d2 t = *reinterpret_cast<const d2 *>(parts + (size_t)p * part_stride + o * ldo + c);
        s.x += t.x;
        s.y += t.y;
    }
    if (ovout) {
        const d2 t = *reinterpret_cast<const d2 *>(ovout + o * ldo + c);
        s.x += t.x;
        s.y += t.y;
    }
    for (uint32_t q = 0; q < rank; q++) {
        const double aq = off_a[o * rank + q];
        s.x += aq * off_w[(size_t)q * ldw + c];
        if (c + 1 < l) s.y += aq * off_w[(size_t)q * ldw + c + 1];
    }
    *reinterpret_cast<d2 *>(out + o * ldo + c) = s;
}

template <int K, int S>
void launch_tile_kernel(Storage &st, const TileArgs &ta, const double *X, uint32_t ldx, uint32_t l, double *parts, uint32_t ldo,
                        uint64_t part_stride, uint32_t n_parts, uint32_t tpp, uint32_t n_items, uint32_t grid) {
    const size_t shmem = (size_t)2 * ((TL_T * ldx * 8 + 1023u) / 1024u) * 1024u;
    if constexpr (S <= 28) {
        SCANRS_HIP(hipFuncSetAttribute((const void *)spmm_tile_kernel_r168<K, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((spmm_tile_kernel_r168<K, S>), dim3(grid), dim3(64 * TL_NW), shmem, st.stream, ta, X, ldx, l, parts, ldo, part_stride,
                           n_parts, tpp, n_items);
    } else {
        SCANRS_HIP(hipFuncSetAttribute((const void *)spmm_tile_kernel<K, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((spmm_tile_kernel<K, S>), dim3(grid), dim3(64 * TL_NW), shmem, st.stream, ta, X, ldx, l, parts, ldo, part_stride,
                           n_parts, tpp, n_items);
    }
}

} // namespace

bool spmm_tiles_ok(const SparseCopy &cp, uint32_t ldx, uint32_t l) {
    return l >= 16 && l <= TL_LMAX && ldx <= TL_LMAX && (ldx & 1u) == 0 && cp.n_outer > 0 && cp.n_inner > 0 && cp.nnz > 0;
}
bool tile_shape_ok(uint32_t K, uint32_t S) { return (K == 4 && (S == 32 || S == 28)) || (K == 3 && S == 32) || (K == 2 && S == 32); }

void launch_gather2d_ov(Storage &st, hipStream_t s, SparseCopy &ov, const double *X, uint32_t ldx, uint32_t l, double *out, uint32_t ldo); // kernels.hip

void launch_spmm_tiles(Storage &st, SparseCopy &cp, const DevMap &map, const double *X, uint32_t ldx, uint32_t l, double *out,
                       uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w, uint32_t ldw) {
    if ((ldx & 1u) || (ldo & 1u)) fail(SCANRS_ERR_ARGUMENT, "panel leading dimensions must be even");
    const uint32_t K = st.tile_k, S = st.tile_s;
    if (!cp.tiles || !cp.tiles->matches(map, K, S)) {
        cp.tiles.reset(); // free the old layout before the new one is allocated
        cp.tiles.reset(tile_layout_build(st, cp, map, K, S), tile_layout_free);
    }
    TileLayout &tl = *cp.tiles;
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t wgg = (uint32_t)((tl.n_groups + TL_NW - 1) / TL_NW);
    // items = (workgroup of groups, part of the tile range), dealt round-robin to one persistent workgroup per CU: about 16
    // rounds, the part count chosen so that the last round is nearly full
    uint32_t parts = std::max<uint32_t>(1u, (uint32_t)(16ull * (uint32_t)n_cu / wgg));
    parts = std::min<uint32_t>(parts, std::max<uint32_t>(1u, tl.n_tiles / 8u));
    const uint32_t tpp = (tl.n_tiles + parts - 1) / parts;
    parts = (tl.n_tiles + tpp - 1) / tpp;
    const uint32_t n_items = wgg * parts;
    const uint32_t grid = std::min<uint32_t>(n_items, (uint32_t)n_cu);
    const uint64_t part_stride = cp.n_outer * (uint64_t)ldo;
    double *pbuf = st.scratch.get<double>("tile_parts", (size_t)parts * part_stride);
    double *ovout = nullptr;
    if (tl.ov.nnz) { // the overflow part through the texture path, beside the tile kernel
        ovout = st.scratch.get<double>("tile_ovout", (size_t)part_stride);
        if (st.tile_overlap) {
            hipStream_t ovs = st.ov();
            SCANRS_HIP(hipEventRecord(st.ev_in, st.stream)); // the panel (and the scratch zero-fills) are ready
            SCANRS_HIP(hipStreamWaitEvent(ovs, st.ev_in, 0));
            launch_gather2d_ov(st, ovs, tl.ov, X, ldx, l, ovout, ldo);
            SCANRS_HIP(hipEventRecord(st.ev_ov, ovs));
        } else {
            launch_gather2d_ov(st, st.stream, tl.ov, X, ldx, l, ovout, ldo);
        }
    }
    TileArgs ta{tl.prow.p, tl.pw.p, tl.n_groups, cp.n_outer, cp.n_inner, tl.n_tiles};
    const double bytes = tl.bytes() + (double)cp.n_inner * l * 8.0 + (double)cp.n_outer * l * 8.0;
    const bool long_outer = cp.n_outer >= cp.n_inner;
    if (st.prof.on)
        st.prof.begin(st.stream, long_outer ? "spmm_tile_kernel/long-outer" : "spmm_tile_kernel/short-outer", bytes,
                      (double)tl.n_groups * tl.n_tiles * S * K * 8.0 * l);
#define SCANRS_TILE(KK, SS) launch_tile_kernel<KK, SS>(st, ta, X, ldx, l, pbuf, ldo, part_stride, parts, tpp, n_items, grid)
    if (K == 4 && S == 32)
        SCANRS_TILE(4, 32);
    else if (K == 4 && S == 28)
        SCANRS_TILE(4, 28);
    else if (K == 3 && S == 32)
        SCANRS_TILE(3, 32);
    else if (K == 2 && S == 32)
        SCANRS_TILE(2, 32);
    else
        fail(SCANRS_ERR_ARGUMENT, "unsupported tile shape K=%u S=%u", K, S);
#undef SCANRS_TILE
    if (st.prof.on) st.prof.end(st.stream);
    if (ovout && st.tile_overlap) SCANRS_HIP(hipStreamWaitEvent(st.stream, st.ev_ov, 0));
    const uint64_t n = cp.n_outer * (uint64_t)((l + 1u) / 2u);
    hipLaunchKernelGGL(tile_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st.stream, pbuf, parts, part_stride, ovout,
                       cp.n_outer, l, ldo, out, off_a, rank, off_w, ldw);
    SCANRS_HIP(hipGetLastError());
}

} // namespace scanrs
