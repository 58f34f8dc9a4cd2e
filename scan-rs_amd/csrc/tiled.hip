// LDS-tiled sparse x dense product for gfx950: the fast path behind launch_spmm_f64.
//
// Why: the plain gather kernel (kernels.hip) reads one 8*l-byte panel row per nonzero from L2 / Infinity
// Cache; that on-chip gather, not HBM, bounds it (~8 TB/s, rocprof r01a). Here the panel is cut into tiles
// of TG consecutive rows that are staged ONCE per workgroup into LDS (two buffers, register-staged
// prefetch), and every nonzero reads its panel row from LDS with one conflict-free ds_read_b128 per wave.
//
// Data layout (built once per orientation, `TileCopy`): nonzeros are regrouped by (16-row group g,
// column tile t, row, column) and packed to 32 bits  [31:24] column in tile | [23:20] row in group |
// [19:0] count.  starts[g*T + t] / lens[g*T + t] locate a wave's segment for a tile.
//
// Kernel shape: workgroup = 16 waves = 256 outer vectors; wave = 16 outer vectors whose 16 x l running
// sums live in registers (lanes own column pairs, so a panel row is one 16-B LDS read per lane); the 64
// nonzeros of a chunk are decoded and mapped one per lane (the MatrixMap chain, one log per lane), then
// consumed with v_readlane broadcasts. One barrier per tile.
#include "common.hpp"

#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace scanrs {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr uint32_t TW = 16;      // waves per workgroup
constexpr uint32_t TR = 16;      // outer vectors per wave
constexpr uint32_t TG_SHIFT = 6; // panel rows per tile = 64
constexpr uint32_t TGS = 1u << TG_SHIFT;
constexpr uint32_t VAL_MASK = 0xFFFFFu;

// duplicated from kernels.hip (kept in one translation unit each so the hot loop inlines)
__device__ __forceinline__ double t_a_ln_a_over_b(double a, double b) { return a == 0.0 ? 0.0 : a * log(a / b); }
__device__ __forceinline__ double t_eval_map(const DevMap &m, uint32_t v, uint32_t outer, uint32_t inner) {
    double x = (double)v;
    for (int i = 0; i < m.n; i++) {
        const DevOp &op = m.ops[i];
        switch (op.kind) {
        case OP_SCALE_AXIS:
            x = op.a[op.a_outer ? outer : inner] * x;
            break;
        case OP_LN_1P:
            x = log(x + 1.0);
            break;
        case OP_LOG2_1P:
            x = log2(x + 1.0);
            break;
        case OP_LOG10_1P:
            x = log10(x + 1.0);
            break;
        case OP_SQUARE:
            x = x * x;
            break;
        case OP_BINOM_DEV: {
            double n = op.a[op.a_outer ? outer : inner], pi = op.b[op.b_outer ? outer : inner];
            double mu = n * pi;
            double d = x - mu;
            double sign = (d != d) ? d : (signbit(d) ? -1.0 : 1.0);
            double inner2 = 2.0 * (t_a_ln_a_over_b(x, mu) + t_a_ln_a_over_b(n - x, n - mu));
            double residual = sign * sqrt(fmax(inner2, 0.0));
            double zero_term = -(sqrt(2.0 * n * log(1.0 / (1.0 - pi))));
            x = residual - zero_term;
            break;
        }
        case OP_BINOM_PEARSON: {
            double n = op.a[op.a_outer ? outer : inner], pi = op.b[op.b_outer ? outer : inner];
            double mu = n * pi;
            double residual = (x - mu) / sqrt(mu * (1.0 - pi));
            double zero_term = -sqrt(n * pi / (1.0 - pi));
            x = residual - zero_term;
            break;
        }
        default:
            break;
        }
    }
    return x;
}

__device__ __forceinline__ uint32_t t_rdlane(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
__device__ __forceinline__ double t_bcast(double v, uint32_t lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, (int)lane);
    hi = __builtin_amdgcn_readlane(hi, (int)lane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------------
// build kernels: wave per work item of the CSR copy
__global__ __launch_bounds__(256) void tile_count_kernel(const uint64_t *__restrict__ indptr,
                                                         const uint32_t *__restrict__ indices,
                                                         const uint32_t *__restrict__ values,
                                                         const Item *__restrict__ items, uint32_t n_items, uint32_t T,
                                                         uint32_t *__restrict__ lens32, uint8_t *__restrict__ cnt8,
                                                         uint32_t *__restrict__ flag) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wid >= n_items) return;
    const Item it = items[wid];
    const uint64_t row_start = indptr[it.row], row_end = indptr[it.row + 1];
    const uint64_t g = it.row / TR;
    const uint32_t row_l = it.row % TR;
    for (uint64_t p = it.start + lane; p < it.start + it.len; p += 64u) {
        const uint32_t col = indices[p];
        const uint32_t t = col >> TG_SHIFT;
        if (values[p] >= VAL_MASK) atomicOr(flag, 1u);
        const bool first = (p == row_start) || ((indices[p - 1] >> TG_SHIFT) != t);
        if (first) {
            // first q in (p, row_end) whose column is in a later tile
            const uint32_t bound = (t + 1u) << TG_SHIFT;
            uint64_t lo = p + 1, hi = row_end < p + TGS ? row_end : p + TGS;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (indices[mid] < bound)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const uint32_t count = (uint32_t)(lo - p);
            cnt8[((g * T + t) << 4) + row_l] = (uint8_t)count;
            atomicAdd(&lens32[g * T + t], count);
        }
    }
}

__global__ __launch_bounds__(256) void tile_fill_kernel(const uint64_t *__restrict__ indptr,
                                                        const uint32_t *__restrict__ indices,
                                                        const uint32_t *__restrict__ values,
                                                        const Item *__restrict__ items, uint32_t n_items, uint32_t T,
                                                        const uint64_t *__restrict__ starts,
                                                        const uint8_t *__restrict__ cnt8, uint32_t *__restrict__ words) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wid >= n_items) return;
    const Item it = items[wid];
    const uint64_t row_start = indptr[it.row];
    const uint64_t g = it.row / TR;
    const uint32_t row_l = it.row % TR;
    for (uint64_t p = it.start + lane; p < it.start + it.len; p += 64u) {
        const uint32_t col = indices[p];
        const uint32_t t = col >> TG_SHIFT;
        // first nonzero of this vector inside tile t: lower_bound(t * TG) in [max(row_start, p - TG + 1), p]
        const uint32_t bound = t << TG_SHIFT;
        uint64_t lo = (p - row_start >= TGS) ? p - TGS + 1 : row_start, hi = p;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (indices[mid] < bound)
                lo = mid + 1;
            else
                hi = mid;
        }
        const uint32_t rank = (uint32_t)(p - lo);
        const uint8_t *c = cnt8 + ((g * T + t) << 4);
        uint32_t pre = 0;
        for (uint32_t r = 0; r < row_l; r++) pre += c[r];
        words[starts[g * T + t] + pre + rank] = ((col & (TGS - 1u)) << 24) | (row_l << 20) | values[p];
    }
}

__global__ void lens_to_u16_kernel(const uint32_t *__restrict__ lens32, uint64_t n, uint16_t *__restrict__ lens) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lens[i] = (uint16_t)lens32[i];
}

// ---------------------------------------------------------------------------------------------------
// the product kernel
__global__ __launch_bounds__(1024) void spmm_tiled_kernel(const uint32_t *__restrict__ words,
                                                           const uint64_t *__restrict__ starts,
                                                           const uint16_t *__restrict__ lens, uint32_t T,
                                                           uint64_t n_outer, uint64_t n_inner, DevMap map,
                                                           const double *__restrict__ X, uint32_t ldx, uint32_t l,
                                                           uint32_t lp, double *__restrict__ out, uint32_t ldo,
                                                           double *__restrict__ slab, uint32_t tiles_per_split,
                                                           const double *__restrict__ off_a, uint32_t rank,
                                                           const double *__restrict__ off_w, uint32_t ldw) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t split = blockIdx.y;
    const uint32_t t0 = split * tiles_per_split;
    const uint32_t t1 = min(T, t0 + tiles_per_split);
    const uint64_t g = (uint64_t)blockIdx.x * TW + w;
    const uint64_t row0 = g * TR;
    const uint64_t *__restrict__ my_starts = starts + g * T;
    const uint16_t *__restrict__ my_lens = lens + g * T;
    const uint32_t colp = lane * 2u;
    const bool act = colp < l;
    const uint32_t row_bytes = lp * 8u;
    const uint32_t buf_bytes = TGS * row_bytes;
    const uint32_t lane_off = lane * 16u;

    d2 acc[TR];
#pragma unroll
    for (uint32_t r = 0; r < TR; r++) acc[r] = (d2){0.0, 0.0};

    // panel tile staging straight into LDS (global_load_lds_dwordx4: no staging registers): wave w copies
    // rows w, w+16, w+32, w+48 of the tile; the LDS row image is lane-linear (lane * 16 B), which is exactly
    // the wave-uniform-base + lane*size destination rule of the instruction.
    auto stage = [&](uint32_t t, uint32_t buf) {
#pragma unroll
        for (uint32_t i = 0; i < TGS / TW; i++) {
            const uint32_t r = w + i * TW;
            const uint64_t grow = ((uint64_t)t << TG_SHIFT) + r;
            if (act && grow < n_inner) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(X + grow * ldx + colp),
                    (__attribute__((address_space(3))) void *)(lds_raw + buf * buf_bytes + r * row_bytes), 16, 0, 0);
            }
        }
    };

    if (t0 < t1) stage(t0, 0);
    __syncthreads();

    // software pipeline over tiles: segment descriptors two tiles ahead, first chunk of words one tile ahead
    uint64_t seg_cur = 0, seg_nxt = 0;
    uint32_t n_cur = 0, n_nxt = 0, word_cur = 0;
    if (t0 < t1) {
        seg_cur = my_starts[t0];
        n_cur = my_lens[t0];
        if (lane < n_cur) word_cur = words[seg_cur + lane];
        if (t0 + 1 < t1) {
            seg_nxt = my_starts[t0 + 1];
            n_nxt = my_lens[t0 + 1];
        }
    }
    for (uint32_t t = t0; t < t1; t++) {
        const uint32_t cur = (t - t0) & 1u;
        if (t + 1 < t1) stage(t + 1, cur ^ 1u);
        uint32_t word_nxt = 0;
        if (t + 1 < t1 && lane < n_nxt) word_nxt = words[seg_nxt + lane];
        uint64_t seg_nn = 0;
        uint32_t n_nn = 0;
        if (t + 2 < t1) {
            seg_nn = my_starts[t + 2];
            n_nn = my_lens[t + 2];
        }
        const unsigned char *__restrict__ tile = lds_raw + cur * buf_bytes + lane_off;

        const uint64_t seg = seg_cur;
        const uint32_t n = n_cur;
        for (uint32_t c = 0; c < n; c += 64u) {
            const bool valid = c + lane < n;
            uint32_t word = word_cur;
            if (c != 0u) {
                word = 0;
                if (valid) word = words[seg + c + lane];
            }
            const uint32_t col_l = word >> 24, row_l = (word >> 20) & 15u, val = word & VAL_MASK;
            double f = 0.0;
            if (valid) f = t_eval_map(map, val, (uint32_t)(row0 + row_l), (t << TG_SHIFT) + col_l);
            const uint32_t addr = col_l * row_bytes;
            uint32_t pos = 0;
#pragma unroll
            for (uint32_t r = 0; r < TR; r++) {
                uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(valid && row_l == r));
                while (cnt >= 2u) {
                    const uint32_t a0 = t_rdlane(addr, pos), a1 = t_rdlane(addr, pos + 1u);
                    const double f0 = t_bcast(f, pos), f1 = t_bcast(f, pos + 1u);
                    if (act) {
                        const d2 x0 = *reinterpret_cast<const d2 *>(tile + a0);
                        const d2 x1 = *reinterpret_cast<const d2 *>(tile + a1);
                        acc[r].x = fma(f0, x0.x, acc[r].x);
                        acc[r].y = fma(f0, x0.y, acc[r].y);
                        acc[r].x = fma(f1, x1.x, acc[r].x);
                        acc[r].y = fma(f1, x1.y, acc[r].y);
                    }
                    pos += 2u;
                    cnt -= 2u;
                }
                if (cnt) {
                    const uint32_t a0 = t_rdlane(addr, pos);
                    const double f0 = t_bcast(f, pos);
                    if (act) {
                        const d2 x0 = *reinterpret_cast<const d2 *>(tile + a0);
                        acc[r].x = fma(f0, x0.x, acc[r].x);
                        acc[r].y = fma(f0, x0.y, acc[r].y);
                    }
                    pos += 1u;
                }
            }
        }
        seg_cur = seg_nxt;
        n_cur = n_nxt;
        word_cur = word_nxt;
        seg_nxt = seg_nn;
        n_nxt = n_nn;
        __syncthreads();
    }

    // epilogue
    const bool direct = gridDim.y == 1;
#pragma unroll
    for (uint32_t r = 0; r < TR; r++) {
        const uint64_t row = row0 + r;
        if (row >= n_outer || !act) continue;
        d2 v = acc[r];
        if (direct) {
            for (uint32_t q = 0; q < rank; q++) { // LowRankOffset: res += u.dot(&v.dot(rhs))
                const double aq = off_a[row * rank + q];
                v.x += aq * off_w[(size_t)q * ldw + colp];
                v.y += aq * off_w[(size_t)q * ldw + colp + 1];
            }
            *reinterpret_cast<d2 *>(out + row * ldo + colp) = v;
        } else {
            *reinterpret_cast<d2 *>(slab + ((size_t)split * n_outer + row) * ldo + colp) = v;
        }
    }
}

// ordered sum over the tile splits (+ offset), one thread per output pair
__global__ __launch_bounds__(256) void tiled_split_reduce_kernel(const double *__restrict__ slab, uint32_t splits,
                                                                 uint64_t n_outer, uint32_t l, double *__restrict__ out,
                                                                 uint32_t ldo, const double *__restrict__ off_a,
                                                                 uint32_t rank, const double *__restrict__ off_w,
                                                                 uint32_t ldw) {
    const uint32_t hp = (l + 1u) / 2u;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_outer * hp) return;
    const uint64_t row = e / hp;
    const uint32_t colp = (uint32_t)(e % hp) * 2u;
    d2 v = (d2){0.0, 0.0};
    for (uint32_t s = 0; s < splits; s++) {
        const d2 x = *reinterpret_cast<const d2 *>(slab + ((size_t)s * n_outer + row) * ldo + colp);
        v.x += x.x;
        v.y += x.y;
    }
    for (uint32_t q = 0; q < rank; q++) {
        const double aq = off_a[row * rank + q];
        v.x += aq * off_w[(size_t)q * ldw + colp];
        v.y += aq * off_w[(size_t)q * ldw + colp + 1];
    }
    *reinterpret_cast<d2 *>(out + row * ldo + colp) = v;
}

// ---------------------------------------------------------------------------------------------------
void build_tile_copy(Storage &st, SparseCopy &cp) {
    TileCopy &tc = cp.tiles;
    tc.tried = true;
    tc.usable = false;
    if (cp.nnz == 0 || cp.n_outer == 0 || cp.n_inner == 0) return;
    const uint64_t n_blocks = (cp.n_outer + TW * TR - 1) / (TW * TR);
    const uint64_t n_groups = n_blocks * TW; // padded: trailing groups are empty
    const uint64_t T = (cp.n_inner + TGS - 1) >> TG_SHIFT;
    const uint64_t n_seg = n_groups * T;
    if (n_seg >= (1ull << 32) || T >= (1ull << 31)) return;
    hipStream_t s = st.stream;
    DevBuf<uint32_t> lens32(n_seg);
    DevBuf<uint8_t> cnt8(n_seg * 16);
    DevBuf<uint32_t> flag(1);
    SCANRS_HIP(hipMemsetAsync(lens32.p, 0, n_seg * 4, s));
    SCANRS_HIP(hipMemsetAsync(cnt8.p, 0, n_seg * 16, s));
    SCANRS_HIP(hipMemsetAsync(flag.p, 0, 4, s));
    const dim3 grid((cp.n_items + 3u) / 4u), block(256);
    hipLaunchKernelGGL(tile_count_kernel, grid, block, 0, s, cp.indptr.p, cp.indices.p, cp.values.p, cp.items.p, cp.n_items,
                       (uint32_t)T, lens32.p, cnt8.p, flag.p);
    uint32_t hflag = 0;
    SCANRS_HIP(hipMemcpyAsync(&hflag, flag.p, 4, hipMemcpyDeviceToHost, s));
    SCANRS_HIP(hipStreamSynchronize(s));
    if (hflag) return; // a count does not fit 20 bits: the gather kernel stays in charge
    tc.starts.alloc(n_seg + 1);
    tc.lens.alloc(n_seg);
    tc.words.alloc(cp.nnz);
    {
        size_t tmp_bytes = 0;
        SCANRS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, lens32.p, tc.starts.p, (uint64_t)0, (size_t)n_seg,
                                           rocprim::plus<uint64_t>(), s));
        DevBuf<char> tmp(std::max<size_t>(tmp_bytes, 16));
        SCANRS_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, lens32.p, tc.starts.p, (uint64_t)0, (size_t)n_seg,
                                           rocprim::plus<uint64_t>(), s));
        SCANRS_HIP(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(lens_to_u16_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, s, lens32.p, n_seg,
                       tc.lens.p);
    hipLaunchKernelGGL(tile_fill_kernel, grid, block, 0, s, cp.indptr.p, cp.indices.p, cp.values.p, cp.items.p, cp.n_items,
                       (uint32_t)T, tc.starts.p, cnt8.p, tc.words.p);
    SCANRS_HIP(hipGetLastError());
    SCANRS_HIP(hipStreamSynchronize(s));
    tc.T = (uint32_t)T;
    tc.n_blocks = (uint32_t)n_blocks;
    tc.usable = true;
}

// out = S X (+ a w) through the tiled path; l <= 128 per pass, wider panels are cut into equal column chunks.
void launch_spmm_tiled(Storage &st, SparseCopy &cp, const DevMap &map, const double *X, uint32_t ldx, uint32_t l,
                       double *out, uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w, uint32_t ldw) {
    const TileCopy &tc = cp.tiles;
    if ((ldx & 1u) || (ldo & 1u)) fail(SCANRS_ERR_ARGUMENT, "panel leading dimensions must be even");
    const uint32_t n_chunks = (l + 127u) / 128u;
    uint32_t lc = (l + n_chunks - 1u) / n_chunks;
    lc = (lc + 1u) & ~1u; // even chunk width keeps every chunk 16-B aligned
    // enough workgroups to fill 256 CUs a few times over: split the tile range when there are few row blocks
    uint32_t splits = 1;
    if (tc.n_blocks < 1024u) splits = std::min<uint32_t>(tc.T, (1024u + tc.n_blocks - 1u) / tc.n_blocks);
    const uint32_t tps = (tc.T + splits - 1u) / splits;
    splits = (tc.T + tps - 1u) / tps;
    double *slab = nullptr;
    if (splits > 1) slab = st.scratch.get<double>("tiled_slab", (size_t)splits * cp.n_outer * ldo);
    static bool lds_attr_set = false;
    if (!lds_attr_set) { // dynamic LDS above 64 KB must be opted into
        SCANRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_tiled_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TGS * 128 * 8));
        lds_attr_set = true;
    }
    for (uint32_t c0 = 0; c0 < l; c0 += lc) {
        const uint32_t lw = std::min(lc, l - c0);
        const uint32_t lp = (lw + 1u) & ~1u;
        const size_t lds_bytes = (size_t)2 * TGS * lp * 8;
        const double bytes = (double)cp.nnz * 8.0 + (double)(cp.n_outer + 1) * 8.0 + (double)cp.n_inner * lw * 8.0 +
                             (double)cp.n_outer * lw * 8.0;
        const double *offw = off_w ? off_w + c0 : nullptr;
        {
            if (st.prof.on) st.prof.begin(st.stream, "spmm_tiled_kernel", bytes);
            hipLaunchKernelGGL(spmm_tiled_kernel, dim3(tc.n_blocks, splits), dim3(TW * 64), lds_bytes, st.stream,
                               tc.words.p, tc.starts.p, tc.lens.p, tc.T, cp.n_outer, cp.n_inner, map, X + c0, ldx, lw, lp,
                               out + c0, ldo, slab ? slab + c0 : nullptr, tps, off_a, rank, offw, ldw);
            if (st.prof.on) st.prof.end(st.stream);
        }
        if (splits > 1) {
            if (st.prof.on) st.prof.begin(st.stream, "tiled_split_reduce", (double)splits * cp.n_outer * lw * 8.0);
            const uint64_t n = cp.n_outer * ((lw + 1u) / 2u);
            hipLaunchKernelGGL(tiled_split_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st.stream, slab + c0,
                               splits, cp.n_outer, lw, out + c0, ldo, off_a, rank, offw, ldw);
            if (st.prof.on) st.prof.end(st.stream);
        }
    }
    SCANRS_HIP(hipGetLastError());
}

} // namespace scanrs
