// HIP kernels for gfx950 (MI355X) + their launchers.
//
// Sparse side (replaces sqz/src/prod.rs and the axis reductions of sqz/src/mat.rs:273-406):
//   spmm_gather_kernel   out[o,:] = sum_j f(v_j, o, i_j) * X[i_j,:]      (CSR kernel prod.rs:123-148;
//                        the CSC scatter prod.rs:190-214 is served by the same gather run on the
//                        transposed copy of the matrix, so no float atomics are needed)
//   row_reduce_kernel    per-outer-vector sum / sum of squares of mapped values
// Dense side (replaces the ndarray / LAPACK calls of scan-rs/src/dim_red/*.rs):
//   gram_kernel          C = X^T Y, tall-skinny, v_mfma_f64_16x16x4_f64
//   gemm_nn_kernel       Out = beta*C + alpha * X W, tall-skinny times small, same MFMA
//
// Wave = 64 lanes everywhere. One wave works one `Item` (<= ITEM_NNZ nonzeros of one outer vector).
#include "common.hpp"
#include "device_map.hpp"

#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace scanrs {

// ---------------------------------------------------------------------------------------------
// Sparse x dense gather product. Lanes own column pairs of the panel (16-B loads, a whole panel row
// is one coalesced read); the 64 nonzeros of a chunk are loaded one per lane, mapped in parallel
// (one log per lane, not per column) and then broadcast one by one with v_readlane.
template <typename T>
struct Vec2;
template <>
struct Vec2<double> {
    typedef d2 type;
};
template <>
struct Vec2<uint32_t> {
    typedef u2 type;
};

template <typename T>
__device__ __forceinline__ T bcast(T v, uint32_t lane);
template <>
__device__ __forceinline__ double bcast<double>(double v, uint32_t lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, (int)lane);
    hi = __builtin_amdgcn_readlane(hi, (int)lane);
    return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ uint32_t bcast<uint32_t>(uint32_t v, uint32_t lane) {
    return rdlane(v, lane);
}

template <typename T>
__device__ __forceinline__ T mul_add(T a, T b, T c);
template <>
__device__ __forceinline__ double mul_add<double>(double a, double b, double c) {
    return fma(a, b, c);
}
template <>
__device__ __forceinline__ uint32_t mul_add<uint32_t>(uint32_t a, uint32_t b, uint32_t c) {
    return c + a * b;
}

template <typename T, int NACC>
__global__ __launch_bounds__(256) void spmm_gather_kernel(const uint32_t *__restrict__ indices,
                                                          const uint32_t *__restrict__ values,
                                                          const Item *__restrict__ items, uint32_t n_items, DevMap map,
                                                          const T *__restrict__ X, uint32_t ldx, uint32_t l,
                                                          T *__restrict__ out, uint32_t ldo, T *__restrict__ slab,
                                                          const double *__restrict__ off_a, uint32_t rank,
                                                          const double *__restrict__ off_w, uint32_t ldw) {
    typedef typename Vec2<T>::type V2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wid >= n_items) return;
    const Item it = items[wid];
    const uint32_t row = rfl(it.row);
    const uint32_t len = rfl(it.len);
    const uint32_t slab_row = rfl(it.slab);
    const uint64_t start = ((uint64_t)rfl((uint32_t)(it.start >> 32)) << 32) | rfl((uint32_t)it.start);
    const uint32_t *__restrict__ ind = indices + start;
    const uint32_t *__restrict__ val = values + start;
    const RowMap rm = row_map(map, row);

    uint32_t col[NACC], lcol[NACC];
    bool act[NACC];
    V2 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        col[a] = (a * 64u + lane) * 2u;
        act[a] = col[a] < l;
        lcol[a] = act[a] ? col[a] : 0u;
        acc[a] = (V2){(T)0, (T)0};
    }

    for (uint32_t base = 0; base < len; base += 64u) {
        const uint32_t p = base + lane;
        uint32_t idx = 0;
        T f = (T)0;
        if (p < len) {
            idx = ind[p];
            if constexpr (sizeof(T) == 8) {
                f = eval_map(map, rm, val[p], row, idx);
            } else {
                f = val[p];
            }
        }
        const uint32_t n = min(64u, len - base);
        uint32_t j = 0;
        for (; j + 8u <= n; j += 8u) {
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t g = rdlane(idx, j + u);
                const T fv = bcast<T>(f, j + u);
                const T *__restrict__ xr = X + (size_t)g * ldx;
#pragma unroll
                for (int a = 0; a < NACC; a++) { // no branch: idle lanes re-read column pair 0 and discard it
                    const V2 x = *reinterpret_cast<const V2 *>(xr + lcol[a]);
                    acc[a].x = mul_add<T>(fv, x.x, acc[a].x);
                    acc[a].y = mul_add<T>(fv, x.y, acc[a].y);
                }
            }
        }
        for (; j < n; j++) {
            const uint32_t g = rdlane(idx, j);
            const T fv = bcast<T>(f, j);
            const T *__restrict__ xr = X + (size_t)g * ldx;
#pragma unroll
            for (int a = 0; a < NACC; a++) { // no branch: idle lanes re-read column pair 0 and discard it
                const V2 x = *reinterpret_cast<const V2 *>(xr + lcol[a]);
                acc[a].x = mul_add<T>(fv, x.x, acc[a].x);
                acc[a].y = mul_add<T>(fv, x.y, acc[a].y);
            }
        }
    }

    if (slab_row == NO_SLAB) {
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            if (act[a]) {
                V2 r = acc[a];
                if constexpr (sizeof(T) == 8) {
                    // LowRankOffset: res += u.dot(&v.dot(rhs))  (sqz/src/low_rank_offset.rs:76-80)
                    for (uint32_t q = 0; q < rank; q++) {
                        const double aq = off_a[(size_t)row * rank + q];
                        r.x += aq * off_w[(size_t)q * ldw + col[a]];
                        r.y += aq * off_w[(size_t)q * ldw + col[a] + 1];
                    }
                }
                *reinterpret_cast<V2 *>(out + (size_t)row * ldo + col[a]) = r;
            }
        }
    } else {
#pragma unroll
        for (int a = 0; a < NACC; a++)
            if (act[a]) *reinterpret_cast<V2 *>(slab + (size_t)slab_row * ldo + col[a]) = acc[a];
    }
}

// L2-blocked variant of the gather product (f64). The panel is cut into steps of `m` base tiles of 1024
// rows, sized so that one step's slice of the panel (<= ~3 MB) stays resident in every XCD's 4 MB L2; one
// launch per step walks ALL outer vectors but only their nonzeros inside the step (bounds table), carrying
// the running sums through `out`. Every wave on the chip then gathers from the same L2-resident slice
// instead of from a 26 MB..4 GB panel spread over Infinity Cache / HBM.
constexpr uint32_t BT_SHIFT = 10;

__global__ void build_bounds_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                    uint64_t n_outer, uint32_t nb, uint32_t *__restrict__ bounds) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_outer * (nb + 1)) return;
    const uint64_t row = e / (nb + 1);
    const uint32_t b = (uint32_t)(e % (nb + 1));
    const uint64_t s = indptr[row], t = indptr[row + 1];
    const uint64_t key = (uint64_t)b << BT_SHIFT;
    uint64_t lo = s, hi = t;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)indices[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    bounds[e] = (uint32_t)(lo - s);
}

// `order` (optional) lists the outer vectors longest first: the launch then starts its longest waves at t = 0
// instead of wherever the storage order puts them. The first `n_hot` vectors of that list (segments of >= ~512
// nonzeros per step: popular genes on the gene-major copy) get a whole workgroup: its 4 waves take a quarter of the
// segment each, partial sums meet in LDS and wave 0 adds them in wave order (deterministic) — otherwise one such
// wave is the critical path of the whole step.
// MAT: the first `fstart` links of the map were evaluated once per nonzero into `fvals` (materialized prefix, see
// ensure_fvals); the kernel reads that f64 instead of the u32 count and applies only the remaining links, which index
// their arrays by the outer position (wave-uniform) — no scattered gather of a scale per nonzero.
template <int NACC, bool MAT>
__global__ __launch_bounds__(256) void spmm_gather2d_kernel(
    const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices, const uint32_t *__restrict__ values,
    const uint32_t *__restrict__ bounds, uint32_t nb, uint32_t b0, uint32_t b1, int first, int last, uint64_t n_outer,
    const uint32_t *__restrict__ order, uint32_t n_hot, DevMap map, const double *__restrict__ X, uint32_t ldx, uint32_t l,
    double *out, uint32_t ldo, const double *__restrict__ off_a, uint32_t rank, const double *__restrict__ off_w, uint32_t ldw,
    const double *__restrict__ fvals, int fstart) {
    __shared__ d2 part[3][NACC][64];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const bool hot = blockIdx.x < n_hot; // block-uniform
    const uint64_t slot = hot ? (uint64_t)blockIdx.x : (uint64_t)n_hot + ((uint64_t)blockIdx.x - n_hot) * 4u + wave;
    if (slot >= n_outer) return; // never taken by a hot block
    const uint64_t row64 = order ? (uint64_t)order[slot] : slot;
    const uint32_t row = (uint32_t)row64;
    const uint32_t *__restrict__ bd = bounds + row64 * (nb + 1);
    uint32_t o0 = rfl(bd[b0]), o1 = rfl(bd[b1]);
    const bool epilogue = last && rank > 0;
    if (!hot && o1 == o0 && !first && !epilogue) return;
    const bool owner = !hot || wave == 0; // holds the carried sums, applies the epilogue, writes the result
    if (hot) {
        uint32_t q = (o1 - o0 + 3u) / 4u;
        q = (q + 7u) & ~7u;
        const uint32_t s0 = min(o1, o0 + wave * q);
        o1 = min(o1, s0 + q);
        o0 = s0;
    }
    const uint32_t len = o1 - o0;
    const uint64_t base = indptr[row64] + o0;
    const uint32_t *__restrict__ ind = indices + base;
    const uint32_t *__restrict__ val = values + base;
    const double *__restrict__ fv = MAT ? fvals + base : nullptr;
    const RowMap rm = row_map(map, row);
    // MAT with a single trailing link (the usual 1/sigma): its factor is one number per outer vector, read once here
    // instead of once per chunk of 64 nonzeros (same product f * a, same rounding)
    const bool one_post = MAT && fstart + 1 == map.n;
    const double post = one_post ? map.ops[fstart & (MAX_OPS - 1)].a[row] : 1.0;

    uint32_t col[NACC], lcol[NACC];
    bool act[NACC];
    d2 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        col[a] = (a * 64u + lane) * 2u;
        act[a] = col[a] < l;
        lcol[a] = act[a] ? col[a] : 0u;
        acc[a] = (d2){0.0, 0.0};
        if (!first && owner && act[a]) acc[a] = *reinterpret_cast<const d2 *>(out + (size_t)row * ldo + col[a]);
    }
    for (uint32_t c = 0; c < len; c += 64u) {
        const uint32_t p = c + lane;
        uint32_t idx = 0;
        double f = 0.0;
        if (p < len) {
            idx = ind[p];
            if constexpr (MAT)
                f = one_post ? post * fv[p] : eval_map_from(map, fstart, fv[p], row, idx);
            else
                f = eval_map(map, rm, val[p], row, idx);
        }
        const uint32_t n = min(64u, len - c);
        // Lanes that own no column sit out the whole gather loop (one exec mask around it, v_readlane still sees
        // their idx / f). Inside, nothing branches per load: with a per-load `if` hipcc puts s_waitcnt vmcnt(0)
        // after every gather and the wave runs latency-bound.
        if (act[0]) {
            uint32_t j = 0;
            for (; j + 8u <= n; j += 8u) {
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) {
                    const uint32_t g = rdlane(idx, j + u);
                    const double fv = bcast<double>(f, j + u);
                    const double *__restrict__ xr = X + (size_t)g * ldx;
#pragma unroll
                    for (int a = 0; a < NACC; a++) { // no branch: lanes idle in slot a re-read column pair 0 and discard it
                        const d2 x = *reinterpret_cast<const d2 *>(xr + lcol[a]);
                        acc[a].x = fma(fv, x.x, acc[a].x);
                        acc[a].y = fma(fv, x.y, acc[a].y);
                    }
                }
            }
            for (; j < n; j++) {
                const uint32_t g = rdlane(idx, j);
                const double fv = bcast<double>(f, j);
                const double *__restrict__ xr = X + (size_t)g * ldx;
#pragma unroll
                for (int a = 0; a < NACC; a++) {
                    const d2 x = *reinterpret_cast<const d2 *>(xr + lcol[a]);
                    acc[a].x = fma(fv, x.x, acc[a].x);
                    acc[a].y = fma(fv, x.y, acc[a].y);
                }
            }
        }
    }
    if (hot) { // block-uniform: all four waves of a hot block reach the barrier
        if (wave > 0) {
#pragma unroll
            for (int a = 0; a < NACC; a++) part[wave - 1][a][lane] = acc[a];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            for (int w = 0; w < 3; w++) {
                const d2 t = part[w][a][lane];
                acc[a].x += t.x;
                acc[a].y += t.y;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        if (!act[a]) continue;
        d2 r = acc[a];
        if (epilogue) {
            for (uint32_t q = 0; q < rank; q++) {
                const double aq = off_a[(size_t)row * rank + q];
                r.x += aq * off_w[(size_t)q * ldw + col[a]];
                r.y += aq * off_w[(size_t)q * ldw + col[a] + 1];
            }
        }
        *reinterpret_cast<d2 *>(out + (size_t)row * ldo + col[a]) = r;
    }
}

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void f64_to_f32_panel_kernel(const double *__restrict__ src, uint32_t lds_, uint64_t rows, uint32_t l,
                                        float *__restrict__ dst, uint32_t ldd) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * ldd) return;
    const uint64_t r = e / ldd;
    const uint32_t c = (uint32_t)(e % ldd);
    dst[e] = c < l ? (float)src[r * lds_ + c] : 0.0f;
}

__global__ __launch_bounds__(256) void spmm_gather2d_f32_kernel(
    const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices, const uint32_t *__restrict__ values,
    const uint32_t *__restrict__ bounds, uint32_t nb, uint32_t b0, uint32_t b1, int first, int last, uint64_t n_outer,
    DevMap map, const float *__restrict__ Xf, uint32_t ldf, uint32_t l, double *out, uint32_t ldo,
    const double *__restrict__ off_a, uint32_t rank, const double *__restrict__ off_w, uint32_t ldw) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t row64 = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row64 >= n_outer) return;
    const uint32_t row = (uint32_t)row64;
    const uint32_t *__restrict__ bd = bounds + row64 * (nb + 1);
    const uint32_t o0 = rfl(bd[b0]), o1 = rfl(bd[b1]);
    const uint32_t len = o1 - o0;
    const bool epilogue = last && rank > 0;
    if (len == 0 && !first && !epilogue) return;
    const uint64_t base = indptr[row64] + o0;
    const uint32_t *__restrict__ ind = indices + base;
    const uint32_t *__restrict__ val = values + base;
    const RowMap rm = row_map(map, row);
    const uint32_t half = lane >> 5, sub = lane & 31u;
    const uint32_t col = sub * 4u;
    const bool act = col < l;
    const uint32_t lcol = act ? col : 0u;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    if (!first && act && half == 0) {
        const double *o = out + (size_t)row * ldo + col;
        acc0 = o[0];
        if (col + 1 < l) acc1 = o[1];
        if (col + 2 < l) acc2 = o[2];
        if (col + 3 < l) acc3 = o[3];
    }
    for (uint32_t c = 0; c < len; c += 64u) {
        const uint32_t p = c + lane;
        uint32_t idx = 0;
        double f = 0.0;
        if (p < len) {
            idx = ind[p];
            f = eval_map(map, rm, val[p], row, idx);
        }
        const uint32_t n = min(64u, len - c);
        // lanes past the chunk hold idx 0 / f 0, so an odd tail simply adds 0 * row 0 in the upper half
        for (uint32_t j = 0; j < n; j += 8u) {
#pragma unroll
            for (uint32_t u = 0; u < 8u; u += 2u) {
                const int srcl = (int)(j + u + half);
                const uint32_t g = (uint32_t)__shfl((int)idx, srcl, 64);
                const double fv = __shfl(f, srcl, 64);
                const f4 x = *reinterpret_cast<const f4 *>(Xf + (size_t)g * ldf + lcol);
                acc0 = fma(fv, (double)x.x, acc0);
                acc1 = fma(fv, (double)x.y, acc1);
                acc2 = fma(fv, (double)x.z, acc2);
                acc3 = fma(fv, (double)x.w, acc3);
            }
        }
    }
    acc0 += __shfl_down(acc0, 32, 64);
    acc1 += __shfl_down(acc1, 32, 64);
    acc2 += __shfl_down(acc2, 32, 64);
    acc3 += __shfl_down(acc3, 32, 64);
    if (half == 0 && act) {
        if (epilogue) {
            for (uint32_t q = 0; q < rank; q++) {
                const double aq = off_a[(size_t)row * rank + q];
                const double *w = off_w + (size_t)q * ldw + col;
                acc0 += aq * w[0];
                if (col + 1 < l) acc1 += aq * w[1];
                if (col + 2 < l) acc2 += aq * w[2];
                if (col + 3 < l) acc3 += aq * w[3];
            }
        }
        double *o = out + (size_t)row * ldo + col;
        o[0] = acc0;
        if (col + 1 < l) o[1] = acc1;
        if (col + 2 < l) o[2] = acc2;
        if (col + 3 < l) o[3] = acc3;
    }
}

// Ordered sum of the partial rows of outer vectors that were cut into several items.
template <typename T, int NACC>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const MultiRow *__restrict__ multi, uint32_t n_multi,
                                                          const T *__restrict__ slab, uint32_t l, T *__restrict__ out,
                                                          uint32_t ldo, const double *__restrict__ off_a, uint32_t rank,
                                                          const double *__restrict__ off_w, uint32_t ldw) {
    typedef typename Vec2<T>::type V2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wid >= n_multi) return;
    const MultiRow mr = multi[wid];
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        const uint32_t col = (a * 64u + lane) * 2u;
        if (col >= l) continue;
        V2 r = (V2){(T)0, (T)0};
        for (uint32_t i = 0; i < mr.count; i++) {
            const V2 x = *reinterpret_cast<const V2 *>(slab + (size_t)(mr.first_slab + i) * ldo + col);
            r.x += x.x;
            r.y += x.y;
        }
        if constexpr (sizeof(T) == 8) {
            for (uint32_t q = 0; q < rank; q++) {
                const double aq = off_a[(size_t)mr.row * rank + q];
                r.x += aq * off_w[(size_t)q * ldw + col];
                r.y += aq * off_w[(size_t)q * ldw + col + 1];
            }
        }
        *reinterpret_cast<V2 *>(out + (size_t)mr.row * ldo + col) = r;
    }
}

// ---------------------------------------------------------------------------------------------
// Axis reductions (sum_axis / mean_var_axis, sqz/src/mat.rs:285-406) as per-outer-vector sums.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}

// Sparse x vector (panel of 1 or 2 columns: the Ix1 `Dot` impls of sqz/src/mat.rs:1092-1112,1150-1170 that
// IRLBA runs on). Lanes stride over the nonzeros of the item, each gathers its own x entries; HBM-bound.
__global__ __launch_bounds__(256) void spmv_kernel(const uint32_t *__restrict__ indices,
                                                   const uint32_t *__restrict__ values,
                                                   const Item *__restrict__ items, uint32_t n_items, DevMap map,
                                                   const double *__restrict__ X, uint32_t ldx, uint32_t l,
                                                   double *__restrict__ out, uint32_t ldo, double *__restrict__ slab,
                                                   const double *__restrict__ off_a, uint32_t rank,
                                                   const double *__restrict__ off_w, uint32_t ldw) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wid >= n_items) return;
    const Item it = items[wid];
    const uint32_t *__restrict__ ind = indices + it.start;
    const uint32_t *__restrict__ val = values + it.start;
    const RowMap rm = row_map(map, it.row);
    double s0 = 0.0, s1 = 0.0;
    // SCAN_U strides of the vector per trip, every load of a trip issued before the first use (clamped position instead
    // of a branch): one round trip to memory per 4 x 64 nonzeros instead of two per 64 — these passes are latency-bound
    for (uint32_t p0 = lane; p0 < it.len; p0 += 64u * SCAN_U) {
        uint32_t g[SCAN_U], vv[SCAN_U];
        bool ok[SCAN_U];
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            const uint32_t p = p0 + 64u * u;
            ok[u] = p < it.len;
            const uint32_t q = ok[u] ? p : it.len - 1u;
            g[u] = ind[q];
            vv[u] = val[q];
        }
        double x0[SCAN_U], x1[SCAN_U];
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            x0[u] = X[(size_t)g[u] * ldx];
            x1[u] = l > 1 ? X[(size_t)g[u] * ldx + 1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            const double f = eval_map(map, rm, vv[u], it.row, g[u]);
            s0 = ok[u] ? fma(f, x0[u], s0) : s0;
            if (l > 1) s1 = ok[u] ? fma(f, x1[u], s1) : s1;
        }
    }
    s0 = wave_sum(s0);
    if (l > 1) s1 = wave_sum(s1);
    if (lane == 0) {
        if (it.slab == NO_SLAB) {
            for (uint32_t q = 0; q < rank; q++) {
                const double aq = off_a[(size_t)it.row * rank + q];
                s0 += aq * off_w[(size_t)q * ldw];
                if (l > 1) s1 += aq * off_w[(size_t)q * ldw + 1];
            }
            out[(size_t)it.row * ldo] = s0;
            if (l > 1) out[(size_t)it.row * ldo + 1] = s1;
        } else {
            slab[(size_t)it.slab * ldo] = s0;
            slab[(size_t)it.slab * ldo + 1] = s1;
        }
    }
}

__global__ void interleave_kernel(const double *__restrict__ x2, const double *__restrict__ a, uint64_t n, double *__restrict__ z) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    z[2 * i] = x2[2 * i]; // the vector sits in slot 0 of its even-ld rows
    z[2 * i + 1] = a[i];
}

// Blocked sparse x vector: the gathered vector (and any inner-indexed scale of the map) is walked in slices that stay
// L2-resident, exactly as row_reduce2d_kernel does for the moments — a straight pass over a gene-major copy turns every
// nonzero into a 64-byte miss on the 8 MB barcode-indexed vector. `order` lists the vectors longest first.
__global__ __launch_bounds__(256) void spmv2d_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                     const uint32_t *__restrict__ values, const uint32_t *__restrict__ bounds,
                                                     uint32_t nb, uint32_t b0, uint32_t b1, int first, int last, uint64_t n_outer,
                                                     const uint32_t *__restrict__ order, DevMap map, int fused_scale,
                                                     const double *__restrict__ X, uint32_t ldx, uint32_t l, double *__restrict__ out,
                                                     uint32_t ldo, const double *__restrict__ off_a, uint32_t rank,
                                                     const double *__restrict__ off_w, uint32_t ldw) {
    // fused_scale (l == 1 only): X is the vector interleaved with the operand of the map's first link, an inner-indexed
    // ScaleAxis — (x[i], a[i]) in the two slots of the even-ld row — so ONE 16-byte gather per nonzero brings both
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t slot = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (slot >= n_outer) return;
    const uint64_t row = order ? (uint64_t)order[slot] : slot;
    const uint32_t *__restrict__ bd = bounds + row * (nb + 1);
    const uint32_t o0 = bd[b0], len = bd[b1] - o0;
    const bool epilogue = last && rank > 0;
    if (len == 0 && !first && !epilogue) return;
    const uint64_t base = indptr[row] + o0;
    const RowMap rm = row_map(map, (uint32_t)row);
    double s0 = 0.0, s1 = 0.0;
    for (uint32_t p0 = lane; p0 < len; p0 += 64u * SCAN_U) {
        uint32_t g[SCAN_U], vv[SCAN_U];
        bool ok[SCAN_U];
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            const uint32_t p = p0 + 64u * u;
            ok[u] = p < len;
            const uint64_t q = base + (ok[u] ? p : len - 1u);
            g[u] = indices[q];
            vv[u] = values[q];
        }
        double x0[SCAN_U], x1[SCAN_U];
        if (fused_scale) {
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {
                const d2 z = *reinterpret_cast<const d2 *>(X + (size_t)g[u] * 2u);
                x0[u] = z.x;
                x1[u] = z.y; // a[inner]
            }
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {
                const double f = eval_map_from(map, 1, x1[u] * (double)vv[u], (uint32_t)row, g[u]);
                s0 = ok[u] ? fma(f, x0[u], s0) : s0;
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            x0[u] = X[(size_t)g[u] * ldx];
            x1[u] = l > 1 ? X[(size_t)g[u] * ldx + 1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            const double f = eval_map(map, rm, vv[u], (uint32_t)row, g[u]);
            s0 = ok[u] ? fma(f, x0[u], s0) : s0;
            if (l > 1) s1 = ok[u] ? fma(f, x1[u], s1) : s1;
        }
    }
    s0 = wave_sum(s0);
    if (l > 1) s1 = wave_sum(s1);
    if (lane == 0) {
        if (!first) {
            s0 += out[row * ldo];
            if (l > 1) s1 += out[row * ldo + 1];
        }
        if (epilogue)
            for (uint32_t q = 0; q < rank; q++) {
                const double aq = off_a[row * rank + q];
                s0 += aq * off_w[(size_t)q * ldw];
                if (l > 1) s1 += aq * off_w[(size_t)q * ldw + 1];
            }
        out[row * ldo] = s0;
        if (l > 1) out[row * ldo + 1] = s1;
    }
}

// Sparse x vector on the LONG side (10^6 cell vectors against a gene-indexed vector of a few hundred KB): the vector is
// too big for one LDS and, gathered from L2, costs one 64-line vector-memory instruction per 64 nonzeros. It is cut into
// <= 144 KB parts (whole bounds tiles); a 16-wave workgroup stages one part in LDS, walks its share of the outer vectors
// through it (ds_read_b64 gathers), and carries the partial sums through `out` between parts.
#ifndef SCANRS_SPMV_SU
#define SCANRS_SPMV_SU 8
#endif
// VM: where a nonzero's value comes from — 0 the lazy chain, 1 the materialized values, 2 the row table. (A template, not a runtime test:
// with the test inside the unrolled load loop the compiler waited for every stride's loads before it issued the next stride's.)
template <int VM>
__global__ __launch_bounds__(1024) void spmv_lds_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                        const uint32_t *__restrict__ values, const uint32_t *__restrict__ bounds,
                                                        uint32_t nb, uint32_t tiles_per_part, uint32_t n_parts, uint64_t n_outer,
                                                        uint64_t rows_per_block, DevMap map, const double *__restrict__ X, uint32_t ldx,
                                                        uint64_t n_inner, double *__restrict__ out, uint32_t ldo,
                                                        const double *__restrict__ off_a, uint32_t rank,
                                                        const double *__restrict__ off_w, uint32_t ldw,
                                                        const double *__restrict__ fvals, int fstart) {
    constexpr bool use_tab = VM == 2;
    constexpr int SU = SCANRS_SPMV_SU; // strides of 64 nonzeros in flight per trip: the walk is bound by the requests it keeps in flight, not by bytes
    extern __shared__ double xs[];
    // use_tab: every link of the map depends on the count and the OUTER position only (what is left of every reference normalisation on
    // the cell-major copy once mat_apply has moved the per-gene scale into the vector): a wave evaluates its row's chain at counts
    // 1 .. 15 once per part and every nonzero is a lookup by its own count — 8 bytes per nonzero (index + count) instead of the
    // 12 of the materialized values, and no 8 bytes per nonzero of them resident (round 6; col_moments_kernel's idea)
    __shared__ double row_tab[16][8][8]; // [wave][row of the batch][count - 1]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_block, r1 = min(n_outer, r0 + rows_per_block);
    const uint32_t part_len = tiles_per_part << BT_SHIFT;
    for (uint32_t p = 0; p < n_parts; p++) {
        const uint64_t g0 = (uint64_t)p * part_len;
        const uint32_t glen = (uint32_t)min((uint64_t)part_len, n_inner - g0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < glen; i += 1024u) xs[i] = X[(g0 + i) * ldx];
        __syncthreads();
        const uint32_t b0 = p * tiles_per_part, b1 = min(nb, b0 + tiles_per_part);
        for (uint64_t row_b = r0 + wave; row_b < r1; row_b += 16u * 8u) {
          if (use_tab) { // the chain at counts 1 .. 8 of the batch's eight rows, one evaluation for all 64 (a table per row and part cost more than the walk)
              const uint64_t trow = row_b + 16u * (lane >> 3);
              row_tab[wave][lane >> 3][lane & 7u] = trow < r1 ? eval_map(map, (lane & 7u) + 1u, (uint32_t)trow, 0u) : 0.0;
          }
          for (uint32_t t = 0; t < 8u; t++) {
            const uint64_t row = row_b + 16u * t;
            if (row >= r1) break;
            const uint32_t *__restrict__ bd = bounds + row * (nb + 1);
            const uint32_t o0 = bd[b0], len = bd[b1] - o0;
            const uint64_t base = indptr[row] + o0;
            const RowMap rm = row_map(map, (uint32_t)row);
            double s0 = 0.0;
            for (uint32_t q0 = lane; q0 < len; q0 += 64u * SU) {
                uint32_t g[SU], vv[SU];
                double fv[SU];
                bool ok[SU];
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const uint32_t q = q0 + 64u * u;
                    ok[u] = q < len;
                    const uint64_t e = base + (ok[u] ? q : len - 1u);
                    g[u] = indices[e];
                    if (VM == 1) // the mapped values, materialized: no logarithm per nonzero per product
                        fv[u] = fvals[e];
                    else
                        vv[u] = values[e];
                }
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    double f;
                    if (VM == 1)
                        f = eval_map_from(map, fstart, fv[u], (uint32_t)row, g[u]);
                    else if (use_tab) {
                        f = row_tab[wave][t][min(vv[u] - 1u, 7u)];
                        // (a wave-uniform branch around the chain: as a select the compiler evaluates the logarithm for every nonzero)
#ifndef SCANRS_SPMV_TAB_NOSLOW
                        if (__builtin_amdgcn_ballot_w64(ok[u] && vv[u] - 1u >= 8u) != 0ull) {
                            if (vv[u] - 1u >= 8u) f = eval_map(map, rm, vv[u], (uint32_t)row, g[u]);
                        }
#endif
                    } else
                        f = eval_map(map, rm, vv[u], (uint32_t)row, g[u]);
                    s0 = ok[u] ? fma(f, xs[g[u] - (uint32_t)g0], s0) : s0;
                }
            }
            s0 = wave_sum(s0);
            if (lane == 0) {
                if (p > 0) s0 += out[row * ldo];
                if (p + 1 == n_parts)
                    for (uint32_t q = 0; q < rank; q++) s0 += off_a[row * rank + q] * off_w[(size_t)q * ldw];
                out[row * ldo] = s0;
            }
          }
        }
    }
}

// Walk of the copy with few, long outer vectors (genes as outer vectors) against arrays indexed by the INNER position
// (the per-barcode scale of normalize(), the vector of an Ix1 product): gathered from L2 every nonzero is its own line
// request — 64 per load instruction, which the texture addresser serves at ~3.6 clk each: 5.9 ms for 10^9 nonzeros however
// few bytes move (moments pass 5.8 ms, gene-major SpMV 6.0 ms, both at 1.4 TB/s). Here the inner dimension is cut into
// slices of whole bounds tiles whose piece of the array fits in LDS; a 16-wave workgroup stages one slice, walks the
// segments of its share of the outer vectors (dealt from the longest-first list) through it with ds_read gathers, and
// leaves one partial per (slice, vector) in a slab that slice_reduce_kernel adds in slice order (deterministic).
//   MODE 1: sums of f and f^2 (moments), optionally storing f per nonzero (the materialized prefix, see ensure_fvals)
//   MODE 0: sum of f * x[inner] (Ix1 product)
// Value of a nonzero: fvals given -> chain from link `fstart` on applied to fvals[p]; else, with `lazy_scale`, the chain's
// first link is an inner-indexed ScaleAxis whose array is staged as well (`sa`); else the plain lazy chain.
// FV: the nonzero's value is read from `fvals` (1) or made from its count (0) — a template parameter for the same reason as in
// spmv_lds_kernel: a runtime test inside the unrolled load loop made the compiler wait for each stride's loads in turn
template <int MODE, int FV>
__global__ __launch_bounds__(1024) void slice_walk_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                          const uint32_t *__restrict__ values, const double *__restrict__ fvals, int fstart,
                                                          const uint32_t *__restrict__ bounds, uint32_t nb, uint32_t tiles_per_slice,
                                                          uint64_t n_outer, uint64_t n_inner, const uint32_t *__restrict__ order,
                                                          uint32_t n_groups, DevMap map, int lazy_scale, const double *__restrict__ X,
                                                          uint32_t ldx, double *__restrict__ slab, double *__restrict__ fout) {
    extern __shared__ double sl[];
    constexpr int SU = MODE == 0 ? SCANRS_SPMV_SU : SCAN_U; // (the product keeps eight strides in flight; the moments pass carries more state per stride)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t slice = blockIdx.x, group = blockIdx.y;
    const uint64_t c0 = (uint64_t)slice * tiles_per_slice << BT_SHIFT;
    const uint32_t slice_len = tiles_per_slice << BT_SHIFT;
    const uint32_t clen = (uint32_t)min((uint64_t)slice_len, n_inner - c0);
    double *__restrict__ sx = sl;                       // MODE 0: x[c0 ..]
    double *__restrict__ sa = MODE == 0 ? sl + slice_len : sl; // the staged scale array (after x for the Ix1 product)
    if (MODE == 0)
        for (uint32_t i = threadIdx.x; i < clen; i += 1024u) sx[i] = X[(c0 + i) * ldx];
    if (lazy_scale) {
        const double *__restrict__ a = map.ops[0].a;
        for (uint32_t i = threadIdx.x; i < clen; i += 1024u) sa[i] = a[c0 + i];
    }
    __syncthreads();
    const uint32_t b0 = slice * tiles_per_slice, b1 = min(nb, b0 + tiles_per_slice);
    const uint32_t cbase = (uint32_t)c0;
    for (uint64_t j = (uint64_t)group + (uint64_t)n_groups * wave; j < n_outer; j += (uint64_t)n_groups * 16u) {
        const uint32_t row = order ? order[j] : (uint32_t)j;
        const uint32_t *__restrict__ bd = bounds + (uint64_t)row * (nb + 1);
        const uint32_t o0 = bd[b0], len = bd[b1] - o0;
        const uint64_t base = indptr[row] + o0;
        const RowMap rm = row_map(map, row);
        double s = 0.0, s2 = 0.0;
        for (uint32_t p0 = lane; p0 < len; p0 += 64u * SU) {
            uint32_t g[SU], vv[SU];
            double fv[SU];
            bool ok[SU];
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const uint32_t p = p0 + 64u * u;
                ok[u] = p < len;
                const uint64_t q = base + (ok[u] ? p : len - 1u);
                g[u] = indices[q];
                if (FV)
                    fv[u] = fvals[q];
                else
                    vv[u] = values[q];
            }
#pragma unroll
            for (int u = 0; u < SU; u++) {
                double f;
                if (FV)
                    f = eval_map_from(map, fstart, fv[u], row, g[u]);
                else if (lazy_scale)
                    f = eval_map_from(map, 1, sa[g[u] - cbase] * (double)vv[u], row, g[u]);
                else
                    f = eval_map(map, rm, vv[u], row, g[u]);
                if constexpr (MODE == 1) {
                    s = ok[u] ? s + f : s;
                    s2 = ok[u] ? fma(f, f, s2) : s2;
                    if (fout && ok[u]) fout[base + p0 + 64u * u] = f;
                } else {
                    s = ok[u] ? fma(f, sx[g[u] - cbase], s) : s;
                }
            }
        }
        s = wave_sum(s);
        if constexpr (MODE == 1) s2 = wave_sum(s2);
        if (lane == 0) {
            if constexpr (MODE == 1) {
                slab[((uint64_t)slice * n_outer + row) * 2u] = s;
                slab[((uint64_t)slice * n_outer + row) * 2u + 1u] = s2;
            } else {
                slab[(uint64_t)slice * n_outer + row] = s;
            }
        }
    }
}
// partials of slice_walk_kernel added in slice order; MODE 0 also applies the rank-r offset of the product
template <int MODE>
__global__ void slice_reduce_kernel(const double *__restrict__ slab, uint32_t n_slices, uint64_t n_outer, double *__restrict__ out_a,
                                    uint32_t ld_a, double *__restrict__ out_b, const double *__restrict__ off_a, uint32_t rank,
                                    const double *__restrict__ off_w, uint32_t ldw) {
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_outer) return;
    if constexpr (MODE == 1) {
        double s = 0.0, s2 = 0.0;
        for (uint32_t k = 0; k < n_slices; k++) {
            s += slab[((uint64_t)k * n_outer + row) * 2u];
            s2 += slab[((uint64_t)k * n_outer + row) * 2u + 1u];
        }
        out_a[row] = s;
        if (out_b) out_b[row] = s2;
    } else {
        double s = 0.0;
        for (uint32_t k = 0; k < n_slices; k++) s += slab[(uint64_t)k * n_outer + row];
        for (uint32_t q = 0; q < rank; q++) s += off_a[row * rank + q] * off_w[(size_t)q * ldw];
        out_a[row * ld_a] = s;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void row_reduce_kernel(const uint32_t *__restrict__ indices,
                                                         const uint32_t *__restrict__ values,
                                                         const Item *__restrict__ items, uint32_t n_items, DevMap map,
                                                         uint32_t *__restrict__ out_u32, double *__restrict__ out_sum,
                                                         double *__restrict__ out_sumsq, uint32_t *__restrict__ slab_u32,
                                                         double *__restrict__ slab_f64) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wid >= n_items) return;
    const Item it = items[wid];
    const uint32_t *__restrict__ ind = indices + it.start;
    const uint32_t *__restrict__ val = values + it.start;
    const RowMap rm = row_map(map, it.row);
    if constexpr (MODE == 0) {
        // 8 independent loads per lane and trip (2 KB per wave in flight): with one the pass held 4.1 TB/s
        uint32_t s = 0;
        for (uint32_t p0 = lane; p0 < it.len; p0 += 64u * 8u) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t p = p0 + 64u * u;
                v[u] = p < it.len ? __builtin_nontemporal_load(val + p) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        s = wave_sum_u32(s);
        if (lane == 0) {
            if (it.slab == NO_SLAB)
                out_u32[it.row] = s;
            else
                slab_u32[it.slab] = s;
        }
    } else {
        double s = 0.0, s2 = 0.0;
        for (uint32_t p0 = lane; p0 < it.len; p0 += 64u * SCAN_U) {
            uint32_t g[SCAN_U], vv[SCAN_U];
            bool ok[SCAN_U];
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {
                const uint32_t p = p0 + 64u * u;
                ok[u] = p < it.len;
                const uint32_t q = ok[u] ? p : it.len - 1u;
                g[u] = ind[q];
                vv[u] = val[q];
            }
#pragma unroll
            for (int u = 0; u < SCAN_U; u++) {
                const double x = eval_map(map, rm, vv[u], it.row, g[u]);
                s = ok[u] ? s + x : s;
                if constexpr (MODE == 2) s2 = ok[u] ? fma(x, x, s2) : s2;
            }
        }
        s = wave_sum(s);
        if constexpr (MODE == 2) s2 = wave_sum(s2);
        if (lane == 0) {
            if (it.slab == NO_SLAB) {
                out_sum[it.row] = s;
                if constexpr (MODE == 2) out_sumsq[it.row] = s2;
            } else {
                slab_f64[2 * (size_t)it.slab] = s;
                if constexpr (MODE == 2) slab_f64[2 * (size_t)it.slab + 1] = s2;
            }
        }
    }
}

// Blocked form of the mapped reductions: when the map reads an array indexed by the INNER position (the per-barcode
// scale while walking gene-major vectors) and that array is larger than an XCD's L2, a straight pass turns every
// nonzero into a 64-B miss (measured 4x the algorithmic HBM bytes). Walking the vectors in steps of inner positions
// whose slice of the array is L2-resident (bounds table) brings the traffic back to the nonzeros themselves.
template <int MODE>
__global__ __launch_bounds__(256) void row_reduce2d_kernel(const uint64_t *__restrict__ indptr,
                                                           const uint32_t *__restrict__ indices,
                                                           const uint32_t *__restrict__ values,
                                                           const uint32_t *__restrict__ bounds, uint32_t nb, uint32_t b0,
                                                           uint32_t b1, int first, uint64_t n_outer, DevMap map,
                                                           double *__restrict__ out_sum, double *__restrict__ out_sumsq,
                                                           double *__restrict__ fout) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t row = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= n_outer) return;
    const uint32_t *__restrict__ bd = bounds + row * (nb + 1);
    const uint32_t o0 = bd[b0], len = bd[b1] - o0;
    if (len == 0 && (!first || MODE == 3)) return;
    const uint64_t base = indptr[row] + o0;
    const RowMap rm = row_map(map, (uint32_t)row);
    double s = 0.0, s2 = 0.0;
    for (uint32_t p0 = lane; p0 < len; p0 += 64u * SCAN_U) {
        uint32_t g[SCAN_U], vv[SCAN_U];
        bool ok[SCAN_U];
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            const uint32_t p = p0 + 64u * u;
            ok[u] = p < len;
            const uint64_t q = base + (ok[u] ? p : len - 1u);
            g[u] = indices[q];
            vv[u] = values[q];
        }
#pragma unroll
        for (int u = 0; u < SCAN_U; u++) {
            const double x = eval_map(map, rm, vv[u], (uint32_t)row, g[u]);
            s = ok[u] ? s + x : s;
            if constexpr (MODE == 2) s2 = ok[u] ? fma(x, x, s2) : s2;
            if (fout && ok[u]) fout[base + p0 + 64u * u] = x; // the mapped value itself, kept for the products (ensure_fvals)
        }
    }
    if constexpr (MODE == 3) return; // values only
    s = wave_sum(s);
    if constexpr (MODE == 2) s2 = wave_sum(s2);
    if (lane == 0) {
        out_sum[row] = first ? s : out_sum[row] + s;
        if constexpr (MODE == 2) out_sumsq[row] = first ? s2 : out_sumsq[row] + s2;
    }
}

template <int MODE>
__global__ void row_reduce_finish_kernel(const MultiRow *__restrict__ multi, uint32_t n_multi,
                                         const uint32_t *__restrict__ slab_u32, const double *__restrict__ slab_f64,
                                         uint32_t *__restrict__ out_u32, double *__restrict__ out_sum,
                                         double *__restrict__ out_sumsq) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_multi) return;
    const MultiRow mr = multi[i];
    if constexpr (MODE == 0) {
        uint32_t s = 0;
        for (uint32_t k = 0; k < mr.count; k++) s += slab_u32[mr.first_slab + k];
        out_u32[mr.row] = s;
    } else {
        double s = 0.0, s2 = 0.0;
        for (uint32_t k = 0; k < mr.count; k++) {
            s += slab_f64[2 * (size_t)(mr.first_slab + k)];
            if constexpr (MODE == 2) s2 += slab_f64[2 * (size_t)(mr.first_slab + k) + 1];
        }
        out_sum[mr.row] = s;
        if constexpr (MODE == 2) out_sumsq[mr.row] = s2;
    }
}

// ---------------------------------------------------------------------------------------------
// w[q,:] = sum_i B[i,q] * X[i,:]   (the `v.dot(rhs)` / `lhs.dot(u)` of low_rank_offset.rs:76-95)
// Streaming form: a wave walks rows r0 + wave, r0 + wave + 4, ... of its block's slice, lanes own column pairs (one 16-byte load
// per lane per row, a panel row = one coalesced read), 8 rows in flight per wave; the four waves' partial sums meet in LDS in
// wave order (deterministic). HBM-bound: n x l x 8 bytes once.
__global__ __launch_bounds__(256) void weighted_colsum_partial_kernel(const double *__restrict__ B, uint32_t rank,
                                                                      const double *__restrict__ X, uint32_t ldx,
                                                                      uint64_t n, uint32_t l, uint64_t rows_per_block,
                                                                      double *__restrict__ partial, double *__restrict__ Xc, uint32_t ldc) {
    __shared__ d2 part[3][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_block;
    const uint64_t r1 = min(n, r0 + rows_per_block);
    for (uint32_t q = 0; q < rank; q++) {
        for (uint32_t c0 = 0; c0 < l; c0 += 128u) {
            const uint32_t col = c0 + lane * 2u;
            const bool act = col < l; // ldx is even and >= l rounded up to even: the pair load stays inside the row
            d2 acc = {0.0, 0.0};
            if (act) {
                uint64_t i = r0 + wave;
                for (; i + 28u < r1; i += 32u) {
                    d2 x[8];
                    double b[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        x[u] = *reinterpret_cast<const d2 *>(X + (i + 4u * u) * ldx + col);
                        b[u] = B[(i + 4u * u) * rank + q];
                    }
                    if (Xc && q == 0u) { // the compact copy of the panel that the tile product stages, from the same read
#pragma unroll
                        for (int u = 0; u < 8; u++) *reinterpret_cast<d2 *>(Xc + (i + 4u * u) * ldc + col) = x[u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        acc.x = fma(b[u], x[u].x, acc.x);
                        acc.y = fma(b[u], x[u].y, acc.y);
                    }
                }
                for (; i < r1; i += 4u) {
                    const d2 x = *reinterpret_cast<const d2 *>(X + i * ldx + col);
                    const double b = B[i * rank + q];
                    if (Xc && q == 0u) *reinterpret_cast<d2 *>(Xc + i * ldc + col) = x;
                    acc.x = fma(b, x.x, acc.x);
                    acc.y = fma(b, x.y, acc.y);
                }
            }
            __syncthreads(); // `part` free again
            if (wave > 0) part[wave - 1][lane] = acc;
            __syncthreads();
            if (wave == 0 && act) {
                for (int w = 0; w < 3; w++) {
                    acc.x += part[w][lane].x;
                    acc.y += part[w][lane].y;
                }
                double *dst = partial + ((size_t)blockIdx.x * rank + q) * l + col;
                dst[0] = acc.x;
                if (col + 1 < l) dst[1] = acc.y;
            }
        }
    }
}
// ordered sum of the per-block partials: 64 output entries per workgroup, the blocks dealt to 4 groups of 64 threads
// (8 independent loads in flight each), the four group sums added in group order (deterministic)
__global__ __launch_bounds__(256) void weighted_colsum_finish_kernel(const double *__restrict__ partial, uint32_t nblocks, uint32_t rank,
                                                                     uint32_t l, double *__restrict__ w, uint32_t ldw) {
    __shared__ double part[4][64];
    const uint32_t e = blockIdx.x * 64u + (threadIdx.x & 63u), g = threadIdx.x >> 6;
    const bool act = e < rank * l;
    double s = 0.0;
    if (act) {
        const size_t stride = (size_t)rank * l;
        uint32_t b = g;
        for (; b + 28u < nblocks; b += 32u) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = partial[(size_t)(b + 4u * u) * stride + e];
#pragma unroll
            for (int u = 0; u < 8; u++) s += t[u];
        }
        for (; b < nblocks; b += 4u) s += partial[(size_t)b * stride + e];
    }
    part[g][threadIdx.x & 63u] = s;
    __syncthreads();
    if (g == 0 && act) {
        const uint32_t q = e / l, c = e % l, t = threadIdx.x;
        w[(size_t)q * ldw + c] = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
    }
}

// ---------------------------------------------------------------------------------------------
// Tall-skinny dense kernels on v_mfma_f64_16x16x4_f64.
// Operand maps (cdna_hip_programming.md §3): lane l gives A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
// the 4 results of lane l are D[row = (l>>4) + 4*reg][col = l&15].

// C = X^T Y over a slice of rows; one wave per (32x32 tile of C, row slice).
__global__ __launch_bounds__(64) void gram_kernel(const double *__restrict__ X, uint32_t ldx, uint32_t n,
                                                  const double *__restrict__ Y, uint32_t ldy, uint32_t m, uint64_t rows,
                                                  uint64_t rows_per_split, uint32_t tiles_m,
                                                  double *__restrict__ slab, const int *__restrict__ skip) {
    if (skip && *skip) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t li = lane & 15u, lk = lane >> 4;
    const uint32_t i0 = (blockIdx.x / tiles_m) * 32u, j0 = (blockIdx.x % tiles_m) * 32u;
    const uint64_t r0 = (uint64_t)blockIdx.y * rows_per_split;
    const uint64_t r1 = min(rows, r0 + rows_per_split);
    const bool ai0 = i0 + li < n, ai1 = i0 + 16u + li < n;
    const bool bj0 = j0 + li < m, bj1 = j0 + 16u + li < m;
    d4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
    for (uint64_t r = r0; r < r1; r += 8u) {
        double a0[2], a1[2], b0[2], b1[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint64_t rr = r + 4u * h + lk;
            const bool v = rr < r1;
            const double *xr = X + rr * ldx + i0 + li;
            const double *yr = Y + rr * ldy + j0 + li;
            a0[h] = (v && ai0) ? xr[0] : 0.0;
            a1[h] = (v && ai1) ? xr[16] : 0.0;
            b0[h] = (v && bj0) ? yr[0] : 0.0;
            b1[h] = (v && bj1) ? yr[16] : 0.0;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h], b0[h], acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h], b1[h], acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h], b0[h], acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h], b1[h], acc11, 0, 0, 0);
        }
    }
    double *__restrict__ dst = slab + (size_t)blockIdx.y * n * m;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const uint32_t ra = i0 + lk + 4u * reg, rb = ra + 16u;
        const uint32_t ca = j0 + li, cb = ca + 16u;
        if (ra < n && ca < m) dst[(size_t)ra * m + ca] = acc00[reg];
        if (ra < n && cb < m) dst[(size_t)ra * m + cb] = acc01[reg];
        if (rb < n && ca < m) dst[(size_t)rb * m + ca] = acc10[reg];
        if (rb < n && cb < m) dst[(size_t)rb * m + cb] = acc11[reg];
    }
}
__global__ void gram_finish_kernel(const double *__restrict__ slab, uint32_t splits, uint64_t nm,
                                   double *__restrict__ C, const int *__restrict__ skip) {
    if (skip && *skip) return;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nm) return;
    double s = 0.0;
    for (uint32_t k = 0; k < splits; k++) s += slab[(size_t)k * nm + e];
    C[e] = s;
}

// Out = beta*Cin + alpha * X W. One wave per 16 rows x 64 columns. The k index is permuted inside a
// 16-deep step (lane group lk owns k0+4*lk .. +3) so that each lane reads 32 contiguous bytes of X.
template <int NJ>
__global__ __launch_bounds__(256) void gemm_nn_kernel(const double *__restrict__ X, uint32_t ldx, uint32_t n,
                                                      const double *__restrict__ W, uint32_t ldw, uint32_t m,
                                                      uint64_t rows, double alpha, double beta,
                                                      const double *Cin, uint32_t ldc, double *Out,
                                                      uint32_t ldo, const int *__restrict__ skip) {
    if (skip && *skip) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t li = lane & 15u, lk = lane >> 4;
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 16u;
    if (r0 >= rows) return;
    const uint32_t j0 = blockIdx.y * 16u * NJ;
    const uint64_t ra = r0 + li;
    const bool rv = ra < rows;
    const double *__restrict__ xrow = X + ra * ldx;
    d4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[j] = (d4){0, 0, 0, 0};
    bool cv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) cv[j] = j0 + 16u * j + li < m;

    for (uint32_t k0 = 0; k0 < n; k0 += 16u) {
        const uint32_t kb = k0 + 4u * lk;
        double xa[4];
        if (rv && kb + 3u < n) {
            const d2 p = *reinterpret_cast<const d2 *>(xrow + kb);
            const d2 q = *reinterpret_cast<const d2 *>(xrow + kb + 2);
            xa[0] = p.x;
            xa[1] = p.y;
            xa[2] = q.x;
            xa[3] = q.y;
        } else {
#pragma unroll
            for (int s = 0; s < 4; s++) xa[s] = (rv && kb + s < n) ? xrow[kb + s] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint32_t kk = kb + s;
            const bool kv = kk < n;
            const double *__restrict__ wr = W + (size_t)kk * ldw + j0 + li;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const double b = (kv && cv[j]) ? wr[16 * j] : 0.0;
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s], b, acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const uint64_t rr = r0 + lk + 4u * reg;
        if (rr >= rows) continue;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            if (!cv[j]) continue;
            const uint32_t cc = j0 + 16u * j + li;
            double r = alpha * acc[j][reg];
            if (beta != 0.0) r = fma(beta, Cin[rr * ldc + cc], r);
            Out[rr * ldo + cc] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// small element-wise helpers
__global__ void copy_cols_kernel(const double *__restrict__ src, uint32_t lds, double *__restrict__ dst, uint32_t ldd,
                                 uint64_t rows, uint32_t l, const int *__restrict__ skip) {
    if (skip && *skip) return;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * l) return;
    const uint64_t r = e / l;
    const uint32_t c = (uint32_t)(e % l);
    dst[r * ldd + c] = src[r * lds + c];
}
// dst[:, j] = src[:, idx[j]] (gather) or dst[:, idx[j]] = src[:, j] (scatter), j < n_idx
__global__ void permute_cols_kernel(const double *__restrict__ src, uint32_t lds, double *__restrict__ dst, uint32_t ldd,
                                    uint64_t rows, const uint32_t *__restrict__ idx, uint32_t n_idx, int scatter) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * n_idx) return;
    const uint64_t r = e / n_idx;
    const uint32_t j = (uint32_t)(e % n_idx);
    if (scatter)
        dst[r * ldd + idx[j]] = src[r * lds + j];
    else
        dst[r * ldd + j] = src[r * lds + idx[j]];
}
// Seeded start panel generated on the device. The reference draws the panel from ONE sequential stream
// (SmallRng = xoshiro256++, scan-rs/src/dim_red/bk_svd.rs:83-90); the state transition of xoshiro is linear over
// GF(2), so the state after s*d draws is J^s * state0 with J = T^d. The host supplies J^(2^k) (solver.cpp); stream s
// rebuilds its start state from the binary digits of s and then draws its d values sequentially. The values and
// their order are identical to the sequential stream.
__device__ __forceinline__ uint64_t xo_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__global__ __launch_bounds__(256) void omega_jump_kernel(const uint64_t *__restrict__ jpow, int n_pow, uint64_t s0, uint64_t s1,
                                                         uint64_t s2, uint64_t s3, uint64_t d, uint64_t total, double *__restrict__ out,
                                                         uint32_t ld, uint64_t seq_cols, int transpose) {
    const uint64_t sid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t e0 = sid * d;
    if (e0 >= total) return;
    uint64_t st[4] = {s0, s1, s2, s3};
    for (int k = 0; k < n_pow; k++) {
        if (!((sid >> k) & 1ull)) continue;
        const uint64_t *__restrict__ m = jpow + (size_t)k * 256 * 4;
        uint64_t nx[4] = {0, 0, 0, 0};
        for (int i = 0; i < 256; i++) {
            const uint64_t *r = m + i * 4;
            const uint64_t x = (r[0] & st[0]) ^ (r[1] & st[1]) ^ (r[2] & st[2]) ^ (r[3] & st[3]);
            nx[i >> 6] |= (uint64_t)(__popcll(x) & 1) << (i & 63);
        }
        st[0] = nx[0];
        st[1] = nx[1];
        st[2] = nx[2];
        st[3] = nx[3];
    }
    const uint64_t e1 = min(total, e0 + d);
    for (uint64_t e = e0; e < e1; e++) {
        const uint64_t result = xo_rotl(st[0] + st[3], 23) + st[0];
        const uint64_t t = st[1] << 17;
        st[2] ^= st[0];
        st[3] ^= st[1];
        st[1] ^= st[2];
        st[0] ^= st[3];
        st[2] ^= t;
        st[3] = xo_rotl(st[3], 45);
        const double v12 = __longlong_as_double((long long)((result >> 12) | 0x3FF0000000000000ull));
        const double v = (v12 - 1.0) * 2.0 + (-1.0);
        // sequential order is row-major over (seq_rows x seq_cols); the device panel is that matrix or its transpose
        const uint64_t i = e / seq_cols, j = e % seq_cols;
        if (transpose)
            out[j * ld + i] = v;
        else
            out[i * ld + j] = v;
    }
}

// dst[j, i] = src[i, j] for src (rows x cols, compact) -> dst (cols x rows, leading dimension ldd); LDS tile transpose
__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ src, uint64_t rows, uint64_t cols,
                                                        double *__restrict__ dst, uint32_t ldd) {
    __shared__ double tile[32][33];
    const uint64_t c0 = (uint64_t)blockIdx.x * 32u, r0 = (uint64_t)blockIdx.y * 32u;
    const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5; // 32 x 8
    for (uint32_t k = ty; k < 32u; k += 8u)
        if (r0 + k < rows && c0 + tx < cols) tile[k][tx] = src[(r0 + k) * cols + c0 + tx];
    __syncthreads();
    for (uint32_t k = ty; k < 32u; k += 8u)
        if (c0 + k < cols && r0 + tx < rows) dst[(c0 + k) * ldd + r0 + tx] = tile[tx][k];
}
__global__ void fill_kernel(double *p, uint64_t n, double v) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) p[e] = v;
}
// scale_and_center finishing step, sqz/src/mat.rs:993-1000 + :966-981 + :946 (neg_means)
__global__ void finish_moments_kernel(const double *__restrict__ sum, const double *__restrict__ sumsq, uint64_t n,
                                      double m, int given_scale, const double *__restrict__ scale_in,
                                      double *__restrict__ neg_mean_over_scale, double *__restrict__ inv_scale,
                                      double *__restrict__ scale_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mean = sum[i] / m;
    double sc;
    if (given_scale) {
        sc = scale_in[i];
    } else {
        const double d = sumsq[i] / m - mean * mean;
        sc = d <= 0.0 ? 1.0 : sqrt(d);
    }
    if (neg_mean_over_scale) neg_mean_over_scale[i] = -(mean / sc);
    if (inv_scale) inv_scale[i] = 1.0 / sc;
    if (scale_out) scale_out[i] = sc;
}
// col_scales = target / count  (scan-rs/src/normalization.rs:169)
__global__ void u32_to_scale_kernel(const uint32_t *__restrict__ counts, uint64_t n, double target,
                                    double *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = target / (double)counts[i];
}
// 12-bit digit histogram of the values matching a prefix: the counting step of the radix select that
// stands in for median_mut's sort (scan-rs/src/stats.rs:13-38).
__global__ __launch_bounds__(256) void hist12_kernel(const uint32_t *__restrict__ v, uint64_t n, uint32_t shift,
                                                     uint32_t digit_mask, uint32_t prefix_mask, uint32_t prefix,
                                                     unsigned long long *__restrict__ hist) {
    __shared__ uint32_t h[4096];
    for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = v[i];
        if ((x & prefix_mask) == prefix) atomicAdd(&h[(x >> shift) & digit_mask], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}
// to_dense (sqz/src/mat.rs:188-205): scatter mapped nonzeros into a zeroed rows_v x cols_v array.
__global__ __launch_bounds__(256) void densify_kernel(const uint32_t *__restrict__ indices,
                                                      const uint32_t *__restrict__ values,
                                                      const Item *__restrict__ items, uint32_t n_items, DevMap map,
                                                      int outer_is_view_row, uint64_t cols_v, double *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wid >= n_items) return;
    const Item it = items[wid];
    for (uint32_t p = lane; p < it.len; p += 64u) {
        const uint32_t in = indices[it.start + p];
        const double x = eval_map(map, values[it.start + p], it.row, in);
        const uint64_t r = outer_is_view_row ? it.row : in, c = outer_is_view_row ? in : it.row;
        out[r * cols_v + c] = x;
    }
}
// pi, u, v of binom_deviance_resid / binom_pearson_resid (scan-rs/src/normalization.rs:232-322)
__global__ void binom_rows_kernel(int kind, const double *__restrict__ rowsum, uint64_t nrows, double total,
                                  double *__restrict__ pi, double *__restrict__ u) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    const double x = rowsum[i] / total;
    pi[i] = x;
    u[i] = kind == OP_BINOM_DEV ? sqrt(log(1.0 / (1.0 - x))) : sqrt(x / (1.0 - x));
}
__global__ void binom_cols_kernel(int kind, const double *__restrict__ n, uint64_t ncols, double *__restrict__ v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols) return;
    v[i] = kind == OP_BINOM_DEV ? -sqrt(2.0 * n[i]) : -sqrt(n[i]);
}
__global__ void sum_f64_kernel(const double *__restrict__ x, uint64_t n, double *__restrict__ out) {
    // single block, fixed order: deterministic total (fit_multinomial_model, normalization.rs:224)
    __shared__ double sh[256];
    double s = 0.0;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// transpose helpers
// packed[p] = (outer id << 32) | count: one wave per work item (a run of at most ITEM_NNZ nonzeros of one outer vector), so the
// outer id is known without a search and the nonzeros stream (a binary search in indptr per nonzero took 17 ms at 10^9)
__global__ __launch_bounds__(256) void pack_outer_kernel(const Item *__restrict__ items, uint32_t n_items, const uint32_t *__restrict__ values,
                                                         unsigned long long *__restrict__ packed) {
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (wid >= n_items) return;
    const Item it = items[wid];
    const unsigned long long hi = (unsigned long long)it.row << 32;
    for (uint32_t k = lane; k < it.len; k += 64u) packed[it.start + k] = hi | values[it.start + k];
}
// One 64-bit sort key per nonzero when (inner index, outer id, count) fit together: key = inner << (ob + cb) | outer << cb | count,
// sorted on the inner bits only (stable: ascending outer ids inside an inner vector are kept). 8 bytes per nonzero and pass instead
// of the 12 of a (u32 key, u64 value) pair sort, and two 8 GB buffers instead of 24 GB of temporaries at 10^9 nonzeros.
__global__ __launch_bounds__(256) void pack_key_kernel(const Item *__restrict__ items, uint32_t n_items, const uint32_t *__restrict__ indices,
                                                       const uint32_t *__restrict__ values, uint32_t ob, uint32_t cb,
                                                       unsigned long long *__restrict__ keys) {
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (wid >= n_items) return;
    const Item it = items[wid];
    const unsigned long long mid = (unsigned long long)it.row << cb;
    for (uint32_t k = lane; k < it.len; k += 64u)
        keys[it.start + k] = ((unsigned long long)indices[it.start + k] << (ob + cb)) | mid | values[it.start + k];
}
__global__ void unpack_key_kernel(const unsigned long long *__restrict__ keys, uint64_t nnz, uint32_t ob, uint32_t cb, uint32_t *__restrict__ ind,
                                  uint32_t *__restrict__ val) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const unsigned long long x = keys[p];
    ind[p] = (uint32_t)((x >> cb) & ((1ull << ob) - 1ull));
    val[p] = (uint32_t)(x & ((1ull << cb) - 1ull));
}
__global__ void lower_bound_key_kernel(const unsigned long long *__restrict__ keys, uint64_t nnz, uint64_t n_inner, uint32_t shift,
                                       uint64_t *__restrict__ indptr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_inner) return;
    uint64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((keys[mid] >> shift) < i)
            lo = mid + 1;
        else
            hi = mid;
    }
    indptr[i] = lo;
}
__global__ void unpack_kernel(const unsigned long long *__restrict__ packed, uint64_t nnz, uint32_t *__restrict__ ind,
                              uint32_t *__restrict__ val) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const unsigned long long x = packed[p];
    ind[p] = (uint32_t)(x >> 32);
    val[p] = (uint32_t)x;
}
__global__ void lower_bound_kernel(const uint32_t *__restrict__ keys, uint64_t nnz, uint64_t n_inner,
                                   uint64_t *__restrict__ indptr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_inner) return;
    uint64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < i)
            lo = mid + 1;
        else
            hi = mid;
    }
    indptr[i] = lo;
}
// Validation as two passes that both stream: (a) over the nonzeros, 4 per thread: stored zeros, indices out of range, and
// DESCENTS — positions whose index is not above its predecessor's, wherever they are; (b) over the outer vectors: a descent at the
// first nonzero of a vector is no violation (it compares with the previous vector's last index) and is taken off again, and
// indptr itself must not decrease. violations = out of range + descents - descents at vector starts + decreasing indptr entries.
// (One thread per outer vector walking its nonzeros, the round-1 form, read every line a dozen times: 45 ms at 10^9 nonzeros.)
__global__ __launch_bounds__(256) void validate_stream_kernel(const uint32_t *__restrict__ indices, const uint32_t *__restrict__ values, uint64_t nnz,
                                                              uint64_t n_inner, unsigned long long *__restrict__ counters) {
    unsigned long long zeros = 0, bad = 0;
    uint32_t vmax = 0;
    const uint4 *I4 = reinterpret_cast<const uint4 *>(indices), *V4 = reinterpret_cast<const uint4 *>(values);
    const uint64_t n4 = (nnz + 3) / 4; // the arrays are padded (DevBuf): the last chunk may be read whole
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n4; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 ix = I4[c], vv = V4[c];
        const uint32_t prev = c ? indices[c * 4 - 1] : 0u;
        const uint32_t i[4] = {ix.x, ix.y, ix.z, ix.w}, v[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t p = c * 4 + k;
            if (p >= nnz) break;
            zeros += v[k] == 0u;
            vmax = v[k] > vmax ? v[k] : vmax;
            bad += (uint64_t)i[k] >= n_inner;
            if (p > 0) bad += i[k] <= (k ? i[k - 1] : prev);
        }
    }
    // wave sums, then one set of atomics per WORKGROUP of a grid of a few thousand (every wave of a 65 536-block grid adding to the
    // same four words was 13 of the pass's 18 ms: same-address atomics are served one after the other)
    for (int off = 32; off > 0; off >>= 1) {
        zeros += __shfl_down(zeros, off);
        bad += __shfl_down(bad, off);
        const uint32_t o = __shfl_down(vmax, off);
        vmax = o > vmax ? o : vmax;
    }
    __shared__ unsigned long long sz[4], sb[4];
    __shared__ uint32_t sm[4];
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) {
        sz[w] = zeros;
        sb[w] = bad;
        sm[w] = vmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long z = sz[0] + sz[1] + sz[2] + sz[3], b = sb[0] + sb[1] + sb[2] + sb[3];
        const uint32_t m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
        if (z) atomicAdd(&counters[0], z);
        if (b) atomicAdd(&counters[1], b);
        atomicMax(&counters[3], (unsigned long long)m); // the largest count: bounds the mapped values (col_moments_kernel) and sizes the transposed copy's sort key
    }
}
__global__ void validate_starts_kernel(const uint64_t *__restrict__ indptr, uint64_t n_outer, const uint32_t *__restrict__ indices, uint64_t nnz,
                                       unsigned long long *__restrict__ counters) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool broken = false, start_descent = false;
    if (o < n_outer) {
        const uint64_t a = indptr[o], b = indptr[o + 1];
        broken = b < a || b > nnz;
        // the first nonzero of a non-empty vector that follows another nonzero: a descent there was counted by the streaming pass
        start_descent = !broken && b > a && a > 0 && indices[a] <= indices[a - 1];
    }
    // nearly EVERY vector start is such a descent (a cell's first gene lies below the previous cell's last one): one atomic per wave,
    // not per vector (10^6 adds to one word took 10 ms)
    const unsigned long long nb = __popcll(__builtin_amdgcn_ballot_w64(broken)), nd = __popcll(__builtin_amdgcn_ballot_w64(start_descent));
    if ((threadIdx.x & 63u) == 0) {
        if (nb) atomicAdd(&counters[1], nb);
        if (nd) atomicAdd(&counters[2], nd);
    }
}

// =============================================================================================
// launchers
// One thread per element. The dispatch packet's grid size is 32 bits of WORK-ITEMS (not blocks), and a launch past it silently runs
// the low 32 bits' worth — refuse instead.
static inline dim3 grid1(uint64_t n, uint32_t block) {
    const uint64_t blocks = (n + block - 1) / block;
    if (blocks * block >= (1ull << 32)) fail(SCANRS_ERR_DEVICE, "a per-element launch over %llu elements exceeds the 2^32 work-items of one dispatch", (unsigned long long)n);
    return dim3((unsigned)blocks);
}

struct ProfScope {
    Storage &st;
    ProfScope(Storage &s, const char *name, double bytes, double onchip = 0.0) : st(s) {
        if (st.prof.on) st.prof.begin(st.stream, name, bytes, onchip);
    }
    ~ProfScope() {
        if (st.prof.on) st.prof.end(st.stream);
    }
};

template <typename T>
static void launch_spmm_t(Storage &st, const SparseCopy &cp, const DevMap &map, const T *X, uint32_t ldx, uint32_t l,
                          T *out, uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w,
                          uint32_t ldw) {
    if (l == 0 || cp.n_outer == 0) return;
    if ((ldx & 1u) || (ldo & 1u)) fail(SCANRS_ERR_ARGUMENT, "panel leading dimensions must be even");
    // column chunks of at most 512 (4 accumulator pairs per lane)
    for (uint32_t c0 = 0; c0 < l; c0 += 512u) {
        const uint32_t lc = std::min(512u, l - c0);
        const uint32_t nacc = (lc + 127u) / 128u;
        T *slab = nullptr;
        if (cp.n_slab) slab = st.scratch.get<T>("spmm_slab", (size_t)cp.n_slab * ldo);
        const dim3 grid((cp.n_items + 3u) / 4u), block(256);
        // algorithmic bytes: nnz*(4+4) + (n_outer+1)*8 + in panel + out panel
        const double bytes = (double)cp.nnz * 8.0 + (double)(cp.n_outer + 1) * 8.0 +
                             (double)cp.n_inner * lc * sizeof(T) + (double)cp.n_outer * lc * sizeof(T);
        const double *offw = off_w ? off_w + c0 : nullptr;
        {
            char pname[48];
            snprintf(pname, sizeof(pname), "spmm_gather_kernel<%s, %u>", sizeof(T) == 8 ? "double" : "unsigned int", nacc);
            ProfScope ps(st, pname, bytes, (double)cp.nnz * sizeof(T) * lc);
#define SCANRS_SPMM(NA)                                                                                              \
    hipLaunchKernelGGL((spmm_gather_kernel<T, NA>), grid, block, 0, st.stream, cp.indices.p, cp.values.p, cp.items.p, \
                       cp.n_items, map, X + c0, ldx, lc, out + c0, ldo, slab ? slab + c0 : nullptr, off_a, rank, offw, \
                       ldw)
            switch (nacc) {
            case 1: SCANRS_SPMM(1); break;
            case 2: SCANRS_SPMM(2); break;
            case 3: SCANRS_SPMM(3); break;
            default: SCANRS_SPMM(4); break;
            }
#undef SCANRS_SPMM
        }
        if (cp.n_multi) {
            ProfScope ps(st, "slab_reduce", (double)cp.n_slab * lc * sizeof(T));
            const dim3 g2((cp.n_multi + 3u) / 4u);
#define SCANRS_SLAB(NA)                                                                                            \
    hipLaunchKernelGGL((slab_reduce_kernel<T, NA>), g2, block, 0, st.stream, cp.multi.p, cp.n_multi, slab + c0, lc, \
                       out + c0, ldo, off_a, rank, offw, ldw)
            switch (nacc) {
            case 1: SCANRS_SLAB(1); break;
            case 2: SCANRS_SLAB(2); break;
            case 3: SCANRS_SLAB(3); break;
            default: SCANRS_SLAB(4); break;
            }
#undef SCANRS_SLAB
        }
    }
    SCANRS_HIP(hipGetLastError());
}

// the table of per-outer-vector offsets at multiples of 1024 inner positions; returns the number of base tiles
static uint32_t ensure_bounds(Storage &st, SparseCopy &cp, hipStream_t s = nullptr) {
    const uint32_t nb = (uint32_t)((cp.n_inner + (1ull << BT_SHIFT) - 1) >> BT_SHIFT);
    if (cp.bounds.n != cp.n_outer * (nb + 1)) {
        cp.bounds.alloc(cp.n_outer * (nb + 1));
        const uint64_t n = cp.n_outer * (nb + 1);
        hipLaunchKernelGGL(build_bounds_kernel, grid1(n, 256), dim3(256), 0, s ? s : st.stream, cp.indptr.p, cp.indices.p, cp.n_outer,
                           nb, cp.bounds.p);
    }
    return nb;
}

__global__ void outer_len_kernel(const uint64_t *__restrict__ indptr, uint64_t n_outer, uint32_t *__restrict__ keys,
                                 uint32_t *__restrict__ ids) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_outer) return;
    const uint64_t len = indptr[r + 1] - indptr[r];
    keys[r] = len > 0xffffffffull ? 0xffffffffu : (uint32_t)len;
    ids[r] = (uint32_t)r;
}

// outer vectors by descending length (stable: equal lengths keep ascending id), lengths mirrored on the host
static void ensure_order(Storage &st, SparseCopy &cp) {
    if (cp.order.n == cp.n_outer && cp.sorted_len.size() == cp.n_outer) return;
    DevBuf<uint32_t> keys_a, keys_b, ids_a;
    keys_a.alloc(cp.n_outer);
    keys_b.alloc(cp.n_outer);
    ids_a.alloc(cp.n_outer);
    cp.order.alloc(cp.n_outer);
    hipLaunchKernelGGL(outer_len_kernel, grid1(cp.n_outer, 256), dim3(256), 0, st.stream, cp.indptr.p, cp.n_outer, keys_a.p, ids_a.p);
    size_t tmp_bytes = 0;
    SCANRS_HIP(rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, keys_a.p, keys_b.p, ids_a.p, cp.order.p, (size_t)cp.n_outer, 0u, 32u,
                                              st.stream));
    DevBuf<unsigned char> tmp;
    tmp.alloc(tmp_bytes ? tmp_bytes : 1);
    SCANRS_HIP(rocprim::radix_sort_pairs_desc(tmp.p, tmp_bytes, keys_a.p, keys_b.p, ids_a.p, cp.order.p, (size_t)cp.n_outer, 0u, 32u,
                                              st.stream));
    cp.sorted_len.resize(cp.n_outer);
    SCANRS_D2H(cp.sorted_len.data(), keys_b.p, cp.n_outer * 4, st.stream);
    SCANRS_SYNC(st.stream);
}

// number of leading vectors of `order` with at least `min_len` nonzeros
static uint32_t count_at_least(const SparseCopy &cp, uint64_t min_len) {
    uint64_t lo = 0, hi = cp.sorted_len.size();
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)cp.sorted_len[mid] >= min_len)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (uint32_t)lo;
}

// ---- materialized prefix of the map chain ---------------------------------------------------------------------------
// On the copy with few, long outer vectors (genes as outer vectors) the chain of normalize() starts with the per-barcode
// scale: an 8-byte read at the INNER index of every nonzero, i.e. one scattered line request per lane per 64 nonzeros,
// in every product (measured: +3.4 ms on a 39.3 ms pass, 1 M x 33 k). The first `n` links — everything up to the trailing
// run of outer-indexed ScaleAxis links — give one f64 per nonzero that does not change between products: it is kept
// (8 B per nonzero beside the 8 B of index + count) and the products apply only the trailing links, which are uniform
// over an outer vector. Identity is by MapOp id (never reused), so a re-normalized handle or another view never sees
// stale values. Same arithmetic in the same order as the lazy evaluation: results are bit-identical.
static int fvals_prefix_len(const DevMap &map) {
    int n = map.n;
    while (n > 0 && map.ops[n - 1].kind == OP_SCALE_AXIS && map.ops[n - 1].a_outer) n--;
    return n;
}
static bool fvals_wanted(Storage &st, const SparseCopy &cp, const DevMap &map, int n) {
    if (!st.materialize || n <= 0 || n > MAX_OPS || cp.n_outer >= cp.n_inner || cp.nnz < st.blocked_min_nnz) return false;
    bool inner_indexed = false;
    for (int i = 0; i < n; i++)
        inner_indexed = inner_indexed || (map.ops[i].a && !map.ops[i].a_outer) || (map.ops[i].b && !map.ops[i].b_outer);
    if (!inner_indexed) return false;
    if (cp.fvals.n == cp.nnz) return true; // already allocated
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    return free_b > cp.nnz * 8 + (size_t(8) << 30); // leave room for the solver's panels
}
// the same store on a copy of any shape when the reason is arithmetic, not a gather: the chain holds a logarithm (or a
// residual map) and the caller is a product that runs many times on a bandwidth budget (the Ix1 products of IRLBA)
static bool fvals_wanted_for_alu(Storage &st, const SparseCopy &cp, const DevMap &map, int n) {
    if (!st.materialize || n <= 0 || n > MAX_OPS || cp.nnz < st.blocked_min_nnz) return false;
    bool heavy = false;
    for (int i = 0; i < n; i++) {
        const int k = map.ops[i].kind;
        heavy = heavy || k == OP_LN_1P || k == OP_LOG2_1P || k == OP_LOG10_1P || k == OP_BINOM_DEV || k == OP_BINOM_PEARSON;
    }
    if (!heavy) return false;
    if (cp.fvals.n == cp.nnz) return true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    return free_b > cp.nnz * 8 + (size_t(8) << 30);
}
static bool fvals_match(const SparseCopy &cp, const DevMap &map, int n) {
    if (cp.fsig_n != n || cp.fvals.n != cp.nnz) return false;
    for (int i = 0; i < n; i++)
        if (cp.fsig_id[i] != map.ops[i].id || cp.fsig_outer[i] != map.ops[i].a_outer) return false;
    return true;
}
// the buffer, about to be (re)written with the values of the first n links of `map`
static double *fvals_claim(Storage &st, SparseCopy &cp, const DevMap &map, int n) {
    if (cp.fvals.n != cp.nnz) cp.fvals.alloc(cp.nnz);
    cp.fsig_n = n;
    for (int i = 0; i < n; i++) {
        cp.fsig_id[i] = map.ops[i].id;
        cp.fsig_outer[i] = map.ops[i].a_outer;
    }
    return cp.fvals.p;
}
// values of the first n links for every nonzero of the copy; computed here when the moments pass did not leave them
static const double *ensure_fvals(Storage &st, SparseCopy &cp, const DevMap &map, int n) {
    if (fvals_match(cp, map, n)) return cp.fvals.p;
    double *fout = fvals_claim(st, cp, map, n);
    DevMap prefix = map;
    prefix.n = n;
    const uint32_t nb = ensure_bounds(st, cp);
    const uint32_t m = 256;
    const uint32_t steps = (nb + m - 1) / m;
    const dim3 grid((unsigned)((cp.n_outer + 3) / 4)), block(256);
    for (uint32_t sidx = 0; sidx < steps; sidx++) {
        const uint32_t b0 = sidx * m, b1 = std::min(nb, b0 + m);
        ProfScope ps(st, "materialize_map_values", (double)cp.nnz * 16.0 / steps);
        hipLaunchKernelGGL((row_reduce2d_kernel<3>), grid, block, 0, st.stream, cp.indptr.p, cp.indices.p, cp.values.p, cp.bounds.p, nb, b0,
                           b1, sidx == 0 ? 1 : 0, cp.n_outer, prefix, (double *)nullptr, (double *)nullptr, fout);
    }
    SCANRS_HIP(hipGetLastError());
    return cp.fvals.p;
}

// ---- slice walk (slice_walk_kernel): eligibility and launch --------------------------------------------------------
// chain shapes the slice walk evaluates: no inner-indexed link at all, or exactly one — a ScaleAxis in first position
static bool slice_walk_chain(const DevMap &map, int from, bool &lazy_scale) {
    lazy_scale = false;
    for (int i = from; i < map.n; i++) {
        const bool inner = (map.ops[i].a && !map.ops[i].a_outer) || (map.ops[i].b && !map.ops[i].b_outer);
        if (!inner) continue;
        if (i == 0 && from == 0 && map.ops[0].kind == OP_SCALE_AXIS && !map.ops[0].a_outer)
            lazy_scale = true;
        else
            return false;
    }
    return true;
}

// MODE 1: out_a = sums, out_b = sums of squares (may be null), fout optional.  MODE 0: out_a[row * ld_a] = product (+ offset)
template <int MODE>
static void launch_slice_walk(Storage &st, SparseCopy &cp, const DevMap &map, const double *fvals, int fstart, bool lazy_scale,
                              const double *X, uint32_t ldx, double *out_a, uint32_t ld_a, double *out_b, double *fout,
                              const double *off_a, uint32_t rank, const double *off_w, uint32_t ldw, const char *label, double bytes) {
    const uint32_t nb = ensure_bounds(st, cp);
    const uint32_t arrays = (MODE == 0 ? 1u : 0u) + (lazy_scale ? 1u : 0u);
    const uint32_t tps = arrays >= 2 ? 8u : 16u; // 8 / 16 tiles of 1024 positions: 128 KB of LDS either way
    const uint32_t n_slices = (nb + tps - 1u) / tps;
    const uint32_t n_groups = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((2048u + n_slices - 1u) / n_slices, (cp.n_outer + 15) / 16));
    const bool ordered = cp.n_outer <= 65536u;
    if (ordered) ensure_order(st, cp);
    const size_t shmem = (size_t)std::max(1u, arrays) * ((size_t)tps << BT_SHIFT) * 8;
    double *slab = st.scratch.get<double>("slice_slab", (size_t)n_slices * cp.n_outer * (MODE == 1 ? 2 : 1));
    // per device, and handles of one process may live on different devices: set on every use (a cheap call)
    ProfScope ps(st, label, bytes);
#define SCANRS_SLICE_WALK(FV)                                                                                                                      \
    do {                                                                                                                                           \
        SCANRS_HIP(hipFuncSetAttribute((const void *)slice_walk_kernel<MODE, FV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));         \
        hipLaunchKernelGGL((slice_walk_kernel<MODE, FV>), dim3(n_slices, n_groups), dim3(1024), shmem, st.stream, cp.indptr.p, cp.indices.p,        \
                           cp.values.p, fvals, fstart, cp.bounds.p, nb, tps, cp.n_outer, cp.n_inner, ordered ? cp.order.p : nullptr, n_groups, map, \
                           lazy_scale ? 1 : 0, X, ldx, slab, fout);                                                                                \
    } while (0)
    if (fvals)
        SCANRS_SLICE_WALK(1);
    else
        SCANRS_SLICE_WALK(0);
#undef SCANRS_SLICE_WALK
    hipLaunchKernelGGL((slice_reduce_kernel<MODE>), grid1(cp.n_outer, 256), dim3(256), 0, st.stream, slab, n_slices, cp.n_outer, out_a, ld_a,
                       out_b, off_a, rank, off_w, ldw);
    SCANRS_HIP(hipGetLastError());
}

// L2-blocked gather: see spmm_gather2d_kernel.
static void launch_spmm_2d(Storage &st, SparseCopy &cp, const DevMap &map, const double *X, uint32_t ldx, uint32_t l,
                           double *out, uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w,
                           uint32_t ldw) {
    if ((ldx & 1u) || (ldo & 1u)) fail(SCANRS_ERR_ARGUMENT, "panel leading dimensions must be even");
    const uint32_t nb = ensure_bounds(st, cp);
    // column chunks of <= 128 keep a panel-row slice <= 1 KB, so a step can hold >= 3 base tiles in L2
    const uint32_t n_chunks = (l + 127u) / 128u;
    uint32_t lc = (l + n_chunks - 1u) / n_chunks;
    lc = (lc + 1u) & ~1u;
    // Longest-first pays when a launch is only a few rounds of waves (33 k genes = 1 round of the chip's 8192 wave slots:
    // -8 % per step); with ~10^6 vectors there is no tail to hide and the permuted rows only scatter the streaming reads (+3 %).
    const bool ordered = st.spmm_order == 2 || (st.spmm_order == 1 && cp.n_outer <= 65536u);
    if (ordered) ensure_order(st, cp);
    const int fstart = fvals_prefix_len(map);
    const double *fvals = fvals_wanted(st, cp, map, fstart) ? ensure_fvals(st, cp, map, fstart) : nullptr;
    const dim3 block(256);
    for (uint32_t c0 = 0; c0 < l; c0 += lc) {
        const uint32_t lw = std::min(lc, l - c0);
        uint32_t m = (uint32_t)(st.l2_tile_bytes / ((size_t)(1u << BT_SHIFT) * lw * 8));
        if (m < 1u) m = 1u;
        const uint32_t steps = (nb + m - 1u) / m;
        // hot = a vector whose average segment per step reaches hot_segment nonzeros
        const uint32_t n_hot = ordered && st.hot_segment ? count_at_least(cp, (uint64_t)st.hot_segment * steps) : 0u;
        const dim3 grid((unsigned)(n_hot + (cp.n_outer - n_hot + 3) / 4));
        const double bytes = ((double)cp.nnz * 8.0 + (double)(cp.n_outer + 1) * 8.0 + (double)cp.n_inner * lw * 8.0 +
                              (double)cp.n_outer * lw * 8.0) / steps;
        const double *offw = off_w ? off_w + c0 : nullptr;
        for (uint32_t sidx = 0; sidx < steps; sidx++) {
            const uint32_t b0 = sidx * m, b1 = std::min(nb, b0 + m);
            ProfScope ps(st, cp.n_outer >= cp.n_inner ? (lw > 110 ? "spmm_gather2d_kernel<1>/long-outer/wide" : "spmm_gather2d_kernel<1>/long-outer")
                                                      : "spmm_gather2d_kernel<1>/short-outer", bytes,
                        (double)cp.nnz * 8.0 * lw / steps);
            if (fvals)
                hipLaunchKernelGGL((spmm_gather2d_kernel<1, true>), grid, block, 0, st.stream, cp.indptr.p, cp.indices.p, cp.values.p,
                                   cp.bounds.p, nb, b0, b1, sidx == 0 ? 1 : 0, sidx + 1 == steps ? 1 : 0, cp.n_outer,
                                   ordered ? cp.order.p : nullptr, n_hot, map, X + c0, ldx, lw, out + c0, ldo, off_a, rank, offw, ldw,
                                   fvals, fstart);
            else
                hipLaunchKernelGGL((spmm_gather2d_kernel<1, false>), grid, block, 0, st.stream, cp.indptr.p, cp.indices.p, cp.values.p,
                                   cp.bounds.p, nb, b0, b1, sidx == 0 ? 1 : 0, sidx + 1 == steps ? 1 : 0, cp.n_outer,
                                   ordered ? cp.order.p : nullptr, n_hot, map, X + c0, ldx, lw, out + c0, ldo, off_a, rank, offw, ldw,
                                   (const double *)nullptr, 0);
        }
    }
    SCANRS_HIP(hipGetLastError());
}

// The overflow part of a tile layout beside the persistent tile kernel (tiles.hip): two gather waves per SIMD, a few
// nonzeros per (vector, step) — the pass is bound by the chain of dependent loads of a task (bounds -> indices / weights ->
// panel rows), not by the texture path. One wave therefore works NV consecutive vectors as ONE task: their bounds in one
// load, their nonzeros (usually fewer than 64 together) in one load of indices and one of weights, the row gathers of all of
// them back to back; the accumulators of the NV vectors are compile-time registers (the gather loop is unrolled over the
// vectors, its bounds are scalars). Weights are materialized (fvals), no map, no offset; sums carried through `out`.
// <= 72 VGPRs: two of these waves fit the registers the two persistent tile waves of a SIMD leave (512 - 2 x 184 = 144).
template <int NV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void spmm_gather_ov_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                             const double *__restrict__ fvals, const uint32_t *__restrict__ bounds, uint32_t nb,
                                                             uint32_t b0, uint32_t b1, int first, uint64_t n_outer, const double *__restrict__ X,
                                                             uint32_t ldx, uint32_t l, double *out, uint32_t ldo) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t task = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t row0 = task * NV;
    if (row0 >= n_outer) return;
    uint32_t len_l = 0, blo = 0, bhi = 0;
    bool any_l = false; // the vector has overflow nonzeros at all: only such rows of `out` are ever written, and tile_finish_kernel reads only those
    if (lane < (uint32_t)NV && row0 + lane < n_outer) {
        const uint32_t *bd = bounds + (row0 + lane) * (nb + 1);
        const uint32_t o0 = bd[b0];
        len_l = bd[b1] - o0;
        const uint64_t i0 = indptr[row0 + lane];
        any_l = indptr[row0 + lane + 1] > i0;
        const uint64_t base = i0 + o0;
        blo = (uint32_t)base;
        bhi = (uint32_t)(base >> 32);
    }
    const uint64_t any_m = __builtin_amdgcn_ballot_w64(any_l);
    uint32_t start[NV + 1];
    uint64_t base[NV];
    start[0] = 0;
#pragma unroll
    for (int r = 0; r < NV; r++) {
        start[r + 1] = start[r] + rdlane(len_l, r);
        base[r] = ((uint64_t)rdlane(bhi, r) << 32) | rdlane(blo, r);
    }
    const uint32_t total = start[NV];
    if (total == 0 && (!first || any_m == 0ull)) return;
    const bool act = lane * 2u < l;
    const uint32_t lcol = act ? lane * 2u : 0u;
    d2 acc[NV];
#pragma unroll
    for (int r = 0; r < NV; r++) {
        acc[r] = (d2){0.0, 0.0};
        if (!first && act && row0 + r < n_outer && start[r + 1] > start[r]) acc[r] = *reinterpret_cast<const d2 *>(out + (row0 + r) * ldo + lcol);
    }
    for (uint32_t c0 = 0; c0 < total; c0 += 64u) {
        const uint32_t p = c0 + lane;
        uint32_t idx = 0;
        double f = 0.0;
        if (p < total) {
            uint64_t a = base[0] + p;
#pragma unroll
            for (int r = 1; r < NV; r++)
                if (p >= start[r]) a = base[r] + (p - start[r]);
            idx = indices[a];
            f = fvals[a];
        }
        if (act) {
#pragma unroll
            for (int r = 0; r < NV; r++) {
                const uint32_t lo = max(start[r], c0) - c0, hi = min(start[r + 1], c0 + 64u);
                if (hi <= c0 + lo) continue; // scalar: nothing of vector r in this chunk
                const uint32_t n = hi - c0;
                uint32_t j = lo;
                for (; j + 8u <= n; j += 8u) {
#pragma unroll
                    for (uint32_t u = 0; u < 8u; u++) {
                        const uint32_t g = rdlane(idx, j + u);
                        const double fv = bcast<double>(f, j + u);
                        const d2 x = *reinterpret_cast<const d2 *>(X + (size_t)g * ldx + lcol);
                        acc[r].x = fma(fv, x.x, acc[r].x);
                        acc[r].y = fma(fv, x.y, acc[r].y);
                    }
                }
                for (; j < n; j++) {
                    const uint32_t g = rdlane(idx, j);
                    const double fv = bcast<double>(f, j);
                    const d2 x = *reinterpret_cast<const d2 *>(X + (size_t)g * ldx + lcol);
                    acc[r].x = fma(fv, x.x, acc[r].x);
                    acc[r].y = fma(fv, x.y, acc[r].y);
                }
            }
        }
    }
    if (act) {
#pragma unroll
        for (int r = 0; r < NV; r++)
            if (row0 + r < n_outer && ((first && ((any_m >> r) & 1ull)) || start[r + 1] > start[r])) *reinterpret_cast<d2 *>(out + (row0 + r) * ldo + lcol) = acc[r];
    }
}

uint32_t ensure_bounds_public(Storage &st, SparseCopy &cp, hipStream_t s) { return ensure_bounds(st, cp, s); }

// fout[p] = the whole chain of `map` at nonzero p of `cp` (tiles.hip: weights of the overflow part)
void materialize_map_values(Storage &st, SparseCopy &cp, const DevMap &map, double *fout) {
    const uint32_t nb = ensure_bounds(st, cp);
    const uint32_t m = 256;
    const uint32_t steps = (nb + m - 1) / m;
    const dim3 grid((unsigned)((cp.n_outer + 3) / 4)), block(256);
    for (uint32_t sidx = 0; sidx < steps; sidx++) {
        const uint32_t b0 = sidx * m, b1 = std::min(nb, b0 + m);
        ProfScope ps(st, "materialize_map_values", (double)cp.nnz * 16.0 / steps);
        hipLaunchKernelGGL((row_reduce2d_kernel<3>), grid, block, 0, st.stream, cp.indptr.p, cp.indices.p, cp.values.p, cp.bounds.p, nb, b0, b1,
                           sidx == 0 ? 1 : 0, cp.n_outer, map, (double *)nullptr, (double *)nullptr, fout);
    }
    SCANRS_HIP(hipGetLastError());
}

// The overflow part of a tile layout (tiles.hip) through the L2-blocked gather on stream `s`: weights are materialized
// (ov.fvals holds the whole chain's value), no vector gets a workgroup, no LDS, the sums start at zero and carry no offset.
void launch_gather2d_ov(Storage &st, hipStream_t s, SparseCopy &ov, const double *X, uint32_t ldx, uint32_t l, double *out,
                        uint32_t ldo) {
    const uint32_t nb = (uint32_t)((ov.n_inner + (1ull << BT_SHIFT) - 1) >> BT_SHIFT); // bounds were built with the layout
    // panel slice per step: twice the L2 slice of the blocked gather unless ov_tile_bytes says otherwise. Beside the tile kernel
    // the gather is bound by the drain at the end of every step, not by L2 hits: 245 steps of 3.5 MB over the 800 MB gene-major
    // panel end 1 ms later than 123 of 7 MB (pass 21.3 vs 20.5 ms; 10.5 MB: 20.8; the whole panel in one step: 24 ms — its
    // Infinity-Cache / HBM row reads slow the tile kernel's staging down). The 26 MB cell-major panel does not care (20.6 ms).
    const size_t ovb = st.ov_tile_bytes ? st.ov_tile_bytes : 2 * st.l2_tile_bytes;
    uint32_t m = (uint32_t)(ovb / ((size_t)(1u << BT_SHIFT) * l * 8));
    if (m < 1u) m = 1u;
    // A small overflow part (the dense tile layout leaves 0.01-3 % of the nonzeros: vectors too sparse to own a slot) is gathered in ONE
    // step over the whole panel: its row reads (below 4 GB) do not disturb the tile kernel's staging, 123 launches of a few
    // thousand nonzeros each took as long as the tile kernel itself (17.5 ms per pass for 0.16 % of the nonzeros).
    if ((double)ov.nnz * l * 8.0 < 4.0e9) m = nb;
    const uint32_t steps = (nb + m - 1u) / m;
    constexpr int NV = 4; // vectors per wave
    const dim3 grid((unsigned)((ov.n_outer + 4u * NV - 1) / (4u * NV))), block(256);
    const double bytes = ((double)ov.nnz * 12.0 + (double)(ov.n_outer + 1) * 8.0 + (double)ov.n_inner * l * 8.0 + (double)ov.n_outer * l * 8.0) / steps;
    for (uint32_t sidx = 0; sidx < steps; sidx++) {
        const uint32_t b0 = sidx * m, b1 = std::min(nb, b0 + m);
        if (st.prof.on) st.prof.begin(s, ov.n_outer >= ov.n_inner ? "spmm_gather2d_ov/long-outer" : "spmm_gather2d_ov/short-outer", bytes, (double)ov.nnz * 8.0 * l / steps);
        hipLaunchKernelGGL((spmm_gather_ov_kernel<NV>), grid, block, 0, s, ov.indptr.p, ov.indices.p, ov.fvals.p, ov.bounds.p, nb, b0, b1,
                           sidx == 0 ? 1 : 0, ov.n_outer, X, ldx, l, out, ldo);
        if (st.prof.on) st.prof.end(s);
    }
    SCANRS_HIP(hipGetLastError());
}

// L2-blocked gather over an f32 copy of the panel (opt-in, see spmm_gather2d_f32_kernel)
static void launch_spmm_2d_f32(Storage &st, SparseCopy &cp, const DevMap &map, const double *X, uint32_t ldx, uint32_t l,
                               double *out, uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w,
                               uint32_t ldw) {
    const uint32_t nb = ensure_bounds(st, cp);
    const uint32_t n_chunks = (l + 127u) / 128u;
    uint32_t lc = (l + n_chunks - 1u) / n_chunks;
    lc = (lc + 3u) & ~3u;
    const dim3 grid((unsigned)((cp.n_outer + 3) / 4)), block(256);
    for (uint32_t c0 = 0; c0 < l; c0 += lc) {
        const uint32_t lw = std::min(lc, l - c0);
        const uint32_t ldf = (lw + 3u) & ~3u;
        float *Xf = st.scratch.get<float>("spmm_xf32", (size_t)cp.n_inner * ldf);
        {
            ProfScope ps(st, "f64_to_f32_panel", (double)cp.n_inner * lw * 12.0);
            hipLaunchKernelGGL(f64_to_f32_panel_kernel, grid1((uint64_t)cp.n_inner * ldf, 256), block, 0, st.stream, X + c0, ldx,
                               cp.n_inner, lw, Xf, ldf);
        }
        uint32_t m = (uint32_t)(st.l2_tile_bytes / ((size_t)(1u << BT_SHIFT) * ldf * 4));
        if (m < 1u) m = 1u;
        const uint32_t steps = (nb + m - 1u) / m;
        const double bytes = ((double)cp.nnz * 8.0 + (double)(cp.n_outer + 1) * 8.0 + (double)cp.n_inner * lw * 4.0 +
                              (double)cp.n_outer * lw * 8.0) / steps;
        const double *offw = off_w ? off_w + c0 : nullptr;
        for (uint32_t sidx = 0; sidx < steps; sidx++) {
            const uint32_t b0 = sidx * m, b1 = std::min(nb, b0 + m);
            ProfScope ps(st, cp.n_outer >= cp.n_inner ? "spmm_gather2d_f32_kernel/long-outer" : "spmm_gather2d_f32_kernel/short-outer",
                         bytes, (double)cp.nnz * 4.0 * lw / steps);
            hipLaunchKernelGGL(spmm_gather2d_f32_kernel, grid, block, 0, st.stream, cp.indptr.p, cp.indices.p, cp.values.p, cp.bounds.p,
                               nb, b0, b1, sidx == 0 ? 1 : 0, sidx + 1 == steps ? 1 : 0, cp.n_outer, map, Xf, ldf, lw, out + c0, ldo,
                               off_a, rank, offw, ldw);
        }
    }
    SCANRS_HIP(hipGetLastError());
}

// Xs[i, :] = a[i] * X[i, :] (a trailing inner-indexed ScaleAxis of the map folded into the gathered panel)
__global__ void scale_rows_kernel(const double *__restrict__ X, uint32_t ldx, uint64_t rows, uint32_t l, const double *__restrict__ a,
                                  double *__restrict__ Xs) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * ldx) return;
    const uint64_t r = e / ldx;
    const uint32_t c = (uint32_t)(e % ldx);
    Xs[e] = c < l ? a[r] * X[e] : 0.0;
}
void launch_scale_rows(Storage &st, const double *X, uint32_t ldx, uint64_t rows, uint32_t l, const double *a, double *Xs) {
    if (rows == 0) return;
    ProfScope ps(st, "scale_rows", (double)rows * l * 16.0);
    hipLaunchKernelGGL(scale_rows_kernel, grid1(rows * ldx, 256), dim3(256), 0, st.stream, X, ldx, rows, l, a, Xs);
    SCANRS_HIP(hipGetLastError());
}

void launch_spmm_f64(Storage &st, SparseCopy &cp, const DevMap &map, const double *X, uint32_t ldx, uint32_t l,
                     double *out, uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w, uint32_t ldw) {
    // hybrid: LDS-staged tiles + gather of the overflow part (tiles.hip). Panels wider than the 104 columns a ring row holds go
    // through it in column chunks of equal width (200 -> 2 x 100, 500 -> 5 x 100): a chunk pass costs what a 100-column pass
    // costs, against two gather instructions per nonzero for every 128 columns of the blocked gather.
    {
        const uint32_t n_chunks = (l + 103u) / 104u;
        const uint32_t lc = n_chunks ? even_up((l + n_chunks - 1u) / n_chunks) : 0u;
        if (l >= 16 && lc >= 16 && spmm_tiles_ok(st, cp, ldx, std::min(lc, l)) && tile_shape_ok(st.tile_k, st.tile_s, st.tile_t, st.tile_b) &&
            (st.spmm_path == 3 || (st.spmm_path == 0 && st.panel_precision == 0 && spmm_tiles_auto(st, cp, map)))) {
            for (uint32_t c0 = 0; c0 < l; c0 += lc) {
                const uint32_t lw = std::min(lc, l - c0);
                if (lw < 16u) { // a last sliver narrower than the tile kernel takes: the plain gather serves it
                    launch_spmm_t<double>(st, cp, map, X + c0, ldx, lw, out + c0, ldo, off_a, rank, off_w ? off_w + c0 : nullptr, ldw);
                    continue;
                }
                launch_spmm_tiles(st, cp, map, X + c0, ldx, lw, out + c0, ldo, off_a, rank, off_w ? off_w + c0 : nullptr, ldw);
            }
            return;
        }
    }
    const bool want_2d = st.spmm_path == 2 || st.spmm_path == 3 || (st.spmm_path == 0 && cp.nnz >= st.blocked_min_nnz && l >= 16);
    if (want_2d && l > 0 && cp.n_outer > 0 && cp.n_inner > 0) {
        if (st.panel_precision == 1)
            launch_spmm_2d_f32(st, cp, map, X, ldx, l, out, ldo, off_a, rank, off_w, ldw);
        else
            launch_spmm_2d(st, cp, map, X, ldx, l, out, ldo, off_a, rank, off_w, ldw);
        return;
    }
    if (l <= 2 && l > 0 && cp.n_outer > 0) {
        if ((ldx & 1u) || (ldo & 1u)) fail(SCANRS_ERR_ARGUMENT, "panel leading dimensions must be even");
        const bool long_outer = cp.n_outer >= cp.n_inner;
        if (st.spmm_path != 1 && st.slice_walk && l == 1 && !long_outer && cp.nnz >= st.blocked_min_nnz && cp.n_inner >= (1ull << 19)) {
            // few long vectors against an inner-indexed vector far beyond L2: slices of it (and of the barcode scale, unless
            // the mapped values are materialized) staged in LDS
            const int fstart = fvals_prefix_len(map);
            bool lazy = false;
            const double *fv = fvals_wanted(st, cp, map, fstart) ? ensure_fvals(st, cp, map, fstart) : nullptr;
            if (fv ? slice_walk_chain(map, fstart, lazy) : slice_walk_chain(map, 0, lazy)) {
                launch_slice_walk<0>(st, cp, map, fv, fstart, lazy, X, ldx, out, ldo, nullptr, nullptr, off_a, rank, off_w, ldw,
                                     "slice_walk_spmv/short-outer",
                                     (double)cp.nnz * (fv ? 12.0 : 8.0) + (double)(cp.n_outer + 1) * 8.0 + (double)(cp.n_inner + cp.n_outer) * 8.0);
                return;
            }
        }
        if (st.spmm_path != 1 && cp.nnz >= st.blocked_min_nnz && cp.n_inner >= (1ull << 19)) {
            // the vector does not fit an XCD's L2: walk it in 2 MB slices (base tiles of 1024 positions x ldx x 8 B)
            const uint32_t nb = ensure_bounds(st, cp);
            const uint32_t m = std::max(1u, 256u / ldx);
            const uint32_t steps = (nb + m - 1u) / m;
            const bool ordered = st.spmm_order == 2 || (st.spmm_order == 1 && cp.n_outer <= 65536u);
            if (ordered) ensure_order(st, cp);
            const dim3 grid((unsigned)((cp.n_outer + 3) / 4)), block(256);
            const double bytes = ((double)cp.nnz * 8.0 + (double)(cp.n_outer + 1) * 8.0 + (double)(cp.n_inner + cp.n_outer) * l * 8.0) / steps;
            // the map's first link reads a[inner] per nonzero (the barcode scale): pair it with x[inner] in one 16-byte slot
            const bool fuse = l == 1 && ldx == 2 && map.n >= 1 && map.ops[0].kind == OP_SCALE_AXIS && !map.ops[0].a_outer;
            const double *Xk = X;
            if (fuse) {
                double *z = st.scratch.get<double>("spmv_z", (size_t)cp.n_inner * 2);
                hipLaunchKernelGGL(interleave_kernel, grid1(cp.n_inner, 256), block, 0, st.stream, X, map.ops[0].a, cp.n_inner, z);
                Xk = z;
            }
            for (uint32_t sidx = 0; sidx < steps; sidx++) {
                const uint32_t b0 = sidx * m, b1 = std::min(nb, b0 + m);
                ProfScope ps(st, long_outer ? "spmv2d_kernel/long-outer" : "spmv2d_kernel/short-outer", bytes);
                hipLaunchKernelGGL(spmv2d_kernel, grid, block, 0, st.stream, cp.indptr.p, cp.indices.p, cp.values.p, cp.bounds.p, nb, b0, b1,
                                   sidx == 0 ? 1 : 0, sidx + 1 == steps ? 1 : 0, cp.n_outer, ordered ? cp.order.p : nullptr, map, fuse ? 1 : 0, Xk,
                                   ldx, l, out, ldo, off_a, rank, off_w, ldw);
            }
            SCANRS_HIP(hipGetLastError());
            return;
        }
        if (st.spmv_lds && l == 1 && st.spmm_path != 1 && cp.nnz >= st.blocked_min_nnz && cp.n_outer >= (1ull << 16) && cp.n_inner >= 8192 &&
            cp.n_inner <= 8u * 18432u) {
            // vector of 64 KB .. 1.1 MB against many outer vectors: LDS-staged parts
            const uint32_t nb = ensure_bounds(st, cp);
            const uint32_t n_parts = (uint32_t)((cp.n_inner + 18431) / 18432);
            const uint32_t tiles_per_part = (uint32_t)(((cp.n_inner + n_parts - 1) / n_parts + (1u << BT_SHIFT) - 1) >> BT_SHIFT);
            const size_t shmem = ((size_t)tiles_per_part << BT_SHIFT) * 8;
            // per device, and handles of one process may live on different devices: set on every use (a cheap call)

            int dev = 0, n_cu = 256;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
            const uint32_t n_blocks = (uint32_t)std::min<uint64_t>((uint64_t)n_cu * 4u, (cp.n_outer + 15) / 16);
            const uint64_t rows_per_block = (cp.n_outer + n_blocks - 1) / n_blocks;
            const int fstart = fvals_prefix_len(map);
            // a chain of the count and the outer position alone: looked up by count from a table the wave makes per row — no materialized values
            bool outer_only = st.spmv_row_table != 0 && map.n > 0 && map_is_simple(map);
            for (int i = 0; i < map.n && outer_only; i++) outer_only = map.ops[i].kind != OP_SCALE_AXIS || map.ops[i].a_outer;
            const double *fv = (!outer_only && fvals_wanted_for_alu(st, cp, map, fstart)) ? ensure_fvals(st, cp, map, fstart) : nullptr;
            ProfScope ps(st, "spmv_lds_kernel/long-outer", (double)cp.nnz * (fv ? 12.0 : 8.0) + (double)(cp.n_outer + 1) * 8.0 + (double)(cp.n_inner + cp.n_outer) * 8.0);
#define SCANRS_SPMV_LDS(VM)                                                                                                                       \
    do {                                                                                                                                          \
        /* per device, and handles of one process may live on different devices: set on every use (a cheap call); beside 8 KB of static LDS */    \
        SCANRS_HIP(hipFuncSetAttribute((const void *)spmv_lds_kernel<VM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));                \
        hipLaunchKernelGGL(spmv_lds_kernel<VM>, dim3(n_blocks), dim3(1024), shmem, st.stream, cp.indptr.p, cp.indices.p, cp.values.p, cp.bounds.p, \
                           nb, tiles_per_part, n_parts, cp.n_outer, rows_per_block, map, X, ldx, cp.n_inner, out, ldo, off_a, rank, off_w, ldw, fv, \
                           fstart);                                                                                                               \
    } while (0)
            if (outer_only)
                SCANRS_SPMV_LDS(2);
            else if (fv)
                SCANRS_SPMV_LDS(1);
            else
                SCANRS_SPMV_LDS(0);
#undef SCANRS_SPMV_LDS
            SCANRS_HIP(hipGetLastError());
            return;
        }
        double *slab = nullptr;
        if (cp.n_slab) slab = st.scratch.get<double>("spmm_slab", (size_t)cp.n_slab * ldo);
        const dim3 grid((cp.n_items + 3u) / 4u), block(256);
        {
            ProfScope ps(st, long_outer ? "spmv_kernel/long-outer" : "spmv_kernel/short-outer", (double)cp.nnz * 8.0 + (double)(cp.n_outer + 1) * 8.0 + (double)(cp.n_inner + cp.n_outer) * l * 8.0);
            hipLaunchKernelGGL(spmv_kernel, grid, block, 0, st.stream, cp.indices.p, cp.values.p, cp.items.p, cp.n_items, map, X, ldx,
                               l, out, ldo, slab, off_a, rank, off_w, ldw);
        }
        if (cp.n_multi)
            hipLaunchKernelGGL((slab_reduce_kernel<double, 1>), dim3((cp.n_multi + 3u) / 4u), block, 0, st.stream, cp.multi.p, cp.n_multi,
                               slab, l, out, ldo, off_a, rank, off_w, ldw);
        SCANRS_HIP(hipGetLastError());
        return;
    }
    launch_spmm_t<double>(st, cp, map, X, ldx, l, out, ldo, off_a, rank, off_w, ldw);
}
void launch_spmm_u32(Storage &st, const SparseCopy &cp, const uint32_t *X, uint32_t ldx, uint32_t l, uint32_t *out,
                     uint32_t ldo) {
    DevMap map;
    memset(&map, 0, sizeof(map));
    launch_spmm_t<uint32_t>(st, cp, map, X, ldx, l, out, ldo, nullptr, 0, nullptr, 0);
}

static uint32_t ensure_bounds(Storage &st, SparseCopy &cp, hipStream_t s);

// Should the moments pass of normalize() keep the mapped values for the gather products? Not when the b-wide products of this
// copy will run through the hybrid tile product, which keeps its own weights: 8 B per nonzero and 1.7 ms of stores per
// normalize saved; a gather product that wants them later (RandSvd's 500-column panels) materializes them itself then.
static bool moments_keep_values(Storage &st, const SparseCopy &cp, const DevMap &map) {
    const bool tiles = st.panel_precision == 0 && tile_shape_ok(st.tile_k, st.tile_s, st.tile_t, st.tile_b) &&
                       (st.spmm_path == 3 || (st.spmm_path == 0 && st.tile_auto && cp.nnz >= std::max<uint64_t>(st.blocked_min_nnz, 1ull << 24)));
    return !tiles && fvals_wanted(st, cp, map, map.n);
}

void launch_row_reduce(Storage &st, SparseCopy &cp, const DevMap &map, int mode, uint32_t *out_u32, double *out_sum,
                       double *out_sumsq) {
    if (cp.n_outer == 0) return;
    if (mode != 0 && cp.nnz >= st.blocked_min_nnz && cp.n_inner >= (1ull << 19)) {
        bool inner_indexed = false;
        for (int i = 0; i < map.n; i++)
            inner_indexed = inner_indexed || (map.ops[i].a && !map.ops[i].a_outer) || (map.ops[i].b && !map.ops[i].b_outer);
        bool lazy = false;
        if (inner_indexed && st.slice_walk && cp.n_outer < cp.n_inner && slice_walk_chain(map, 0, lazy) && lazy) {
            // the barcode scale staged in LDS slice by slice instead of gathered from L2 (see slice_walk_kernel)
            double *fout = mode == 2 && moments_keep_values(st, cp, map) ? fvals_claim(st, cp, map, map.n) : nullptr;
            launch_slice_walk<1>(st, cp, map, nullptr, 0, true, nullptr, 0, out_sum, 1, mode == 2 ? out_sumsq : nullptr, fout, nullptr, 0,
                                 nullptr, 0, mode == 1 ? "slice_walk_sum" : "slice_walk_moments",
                                 (double)cp.nnz * (fout ? 16.0 : 8.0) + (double)(cp.n_outer + 1) * 8.0 + (double)cp.n_outer * 16.0);
            return;
        }
        if (inner_indexed) {
            const uint32_t nb = ensure_bounds(st, cp);
            const uint32_t m = 256; // 256 * 1024 inner positions * 8 B = 2 MB slice of the scale array
            const uint32_t steps = (nb + m - 1) / m;
            const dim3 grid((unsigned)((cp.n_outer + 3) / 4)), block(256);
            // the moments walk evaluates exactly the chain the later products start with (normalize: scale, log — then
            // the 1/sigma link is appended): keep the values, the products then skip the per-nonzero scale gather
            double *fout = mode == 2 && moments_keep_values(st, cp, map) ? fvals_claim(st, cp, map, map.n) : nullptr;
            const double bytes = ((double)cp.nnz * (fout ? 16.0 : 8.0) + (double)(cp.n_outer + 1) * 8.0 + (double)cp.n_outer * (mode == 1 ? 8.0 : 16.0)) / steps;
            for (uint32_t sidx = 0; sidx < steps; sidx++) {
                const uint32_t b0 = sidx * m, b1 = std::min(nb, b0 + m);
                ProfScope ps(st, mode == 1 ? "row_reduce2d_sum" : "row_reduce2d_moments", bytes);
                if (mode == 1)
                    hipLaunchKernelGGL((row_reduce2d_kernel<1>), grid, block, 0, st.stream, cp.indptr.p, cp.indices.p, cp.values.p,
                                       cp.bounds.p, nb, b0, b1, sidx == 0 ? 1 : 0, cp.n_outer, map, out_sum, out_sumsq, (double *)nullptr);
                else
                    hipLaunchKernelGGL((row_reduce2d_kernel<2>), grid, block, 0, st.stream, cp.indptr.p, cp.indices.p, cp.values.p,
                                       cp.bounds.p, nb, b0, b1, sidx == 0 ? 1 : 0, cp.n_outer, map, out_sum, out_sumsq, fout);
            }
            SCANRS_HIP(hipGetLastError());
            return;
        }
    }
    uint32_t *slab_u32 = nullptr;
    double *slab_f64 = nullptr;
    if (cp.n_slab) {
        slab_u32 = st.scratch.get<uint32_t>("rr_slab_u32", cp.n_slab);
        slab_f64 = st.scratch.get<double>("rr_slab_f64", 2 * (size_t)cp.n_slab);
    }
    const dim3 grid((cp.n_items + 3u) / 4u), block(256);
    const double bytes = (double)cp.nnz * (mode == 0 ? 4.0 : 8.0) + (double)(cp.n_outer + 1) * 8.0 +
                         (double)cp.n_outer * (mode == 0 ? 4.0 : (mode == 1 ? 8.0 : 16.0));
    {
        ProfScope ps(st, mode == 0 ? "row_reduce_u32" : (mode == 1 ? "row_reduce_sum" : "row_reduce_moments"), bytes);
        if (mode == 0)
            hipLaunchKernelGGL((row_reduce_kernel<0>), grid, block, 0, st.stream, cp.indices.p, cp.values.p, cp.items.p,
                               cp.n_items, map, out_u32, out_sum, out_sumsq, slab_u32, slab_f64);
        else if (mode == 1)
            hipLaunchKernelGGL((row_reduce_kernel<1>), grid, block, 0, st.stream, cp.indices.p, cp.values.p, cp.items.p,
                               cp.n_items, map, out_u32, out_sum, out_sumsq, slab_u32, slab_f64);
        else
            hipLaunchKernelGGL((row_reduce_kernel<2>), grid, block, 0, st.stream, cp.indices.p, cp.values.p, cp.items.p,
                               cp.n_items, map, out_u32, out_sum, out_sumsq, slab_u32, slab_f64);
    }
    if (cp.n_multi) {
        const dim3 g2 = grid1(cp.n_multi, 256);
        if (mode == 0)
            hipLaunchKernelGGL((row_reduce_finish_kernel<0>), g2, block, 0, st.stream, cp.multi.p, cp.n_multi, slab_u32,
                               slab_f64, out_u32, out_sum, out_sumsq);
        else if (mode == 1)
            hipLaunchKernelGGL((row_reduce_finish_kernel<1>), g2, block, 0, st.stream, cp.multi.p, cp.n_multi, slab_u32,
                               slab_f64, out_u32, out_sum, out_sumsq);
        else
            hipLaunchKernelGGL((row_reduce_finish_kernel<2>), g2, block, 0, st.stream, cp.multi.p, cp.n_multi, slab_u32,
                               slab_f64, out_u32, out_sum, out_sumsq);
    }
    SCANRS_HIP(hipGetLastError());
}

// Xc (optional): a compact copy of X's first l columns in rows of ldc, written from the same read (l even: the lanes move column pairs)
// C = X^T y for ONE vector y (rows x 1, stride ldy) and up to 128 columns of X: IRLBA's re-orthogonalisation dots (irlba.rs:19-22,
// 131-134, 10^6 x <= 70 against one Lanczos vector). The 32 x 32 MFMA tiles of gram_kernel spend 31 of 32 columns on padding there and
// load 16 lanes x 8 B per row (0.6-0.8 ms per call at 10^6 rows); here a lane owns columns lane and lane + 64, a wave walks rows
// r0 + wave, + 4, ... of its block's slice (a row of X = one coalesced read, y[r] one broadcast load), eight rows in flight; the four
// waves' sums meet in LDS in wave order, the blocks' partials are added in block order by gram_finish_kernel: deterministic.
__global__ __launch_bounds__(256) void gram_vec_kernel(const double *__restrict__ X, uint32_t ldx, uint32_t n, const double *__restrict__ y, uint32_t ldy,
                                                       uint64_t rows, uint64_t rows_per_block, double *__restrict__ slab, const int *__restrict__ skip) {
    if (skip && *skip) return;
    __shared__ double part[3][128];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    const bool a0 = lane < n, a1 = lane + 64u < n;
    double acc0 = 0.0, acc1 = 0.0;
    uint64_t i = r0 + wave;
    for (; i + 28u < r1; i += 32u) {
        double x0[8], x1[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double *xr = X + (i + 4u * u) * ldx;
            x0[u] = a0 ? xr[lane] : 0.0;
            x1[u] = a1 ? xr[lane + 64u] : 0.0;
            yv[u] = y[(i + 4u * u) * ldy];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc0 = fma(x0[u], yv[u], acc0);
            acc1 = fma(x1[u], yv[u], acc1);
        }
    }
    for (; i < r1; i += 4u) {
        const double *xr = X + i * ldx;
        const double yv = y[i * ldy];
        acc0 = fma(a0 ? xr[lane] : 0.0, yv, acc0);
        acc1 = fma(a1 ? xr[lane + 64u] : 0.0, yv, acc1);
    }
    if (wave > 0) {
        part[wave - 1][lane] = acc0;
        part[wave - 1][lane + 64u] = acc1;
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < 3; w++) {
            acc0 += part[w][lane];
            acc1 += part[w][lane + 64u];
        }
        double *dst = slab + (size_t)blockIdx.x * n;
        if (a0) dst[lane] = acc0;
        if (a1) dst[lane + 64u] = acc1;
    }
}

void launch_weighted_colsum(Storage &st, const double *B, uint32_t rank, const double *X, uint32_t ldx, uint64_t n,
                            uint32_t l, double *w, uint32_t ldw, double *Xc, uint32_t ldc) {
    if (rank == 0 || l == 0) return;
    uint32_t nblocks = (uint32_t)std::min<uint64_t>(1024, (n + 255) / 256);
    if (nblocks == 0) nblocks = 1;
    const uint64_t rpb = (n + nblocks - 1) / nblocks;
    double *partial = st.scratch.get<double>("wcs_partial", (size_t)nblocks * rank * l);
    ProfScope ps(st, "weighted_colsum", (double)n * l * 8.0 + (double)n * rank * 8.0);
    hipLaunchKernelGGL(weighted_colsum_partial_kernel, dim3(nblocks), dim3(256), 0, st.stream, B, rank, X, ldx, n, l, rpb,
                       partial, Xc, ldc);
    hipLaunchKernelGGL(weighted_colsum_finish_kernel, grid1((uint64_t)rank * l, 64), dim3(256), 0, st.stream, partial,
                       nblocks, rank, l, w, ldw);
    SCANRS_HIP(hipGetLastError());
}

void launch_gram(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *Y, uint32_t ldy, uint32_t m,
                 uint64_t rows, double *C) {
    if (n == 0 || m == 0) return;
    if (m == 1u && n <= 128u && rows >= 4096u && !st.skip_flag) { // one vector against a narrow panel: the streaming form (gram_vec_kernel)
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(1024, (rows + 255) / 256);
        const uint64_t rpb = (rows + blocks - 1) / blocks;
        double *slab = st.scratch.get<double>(st.skey("gram_slab"), (size_t)blocks * n);
        ProfScope ps(st, "gram_vec_f64", (double)rows * (n + 1) * 8.0);
        hipLaunchKernelGGL(gram_vec_kernel, dim3(blocks), dim3(256), 0, st.stream, X, ldx, n, Y, ldy, rows, rpb, slab, st.skip_flag);
        // (the blocks' partials in block order, four groups of 64 threads with eight loads in flight each: the plain ordered loop of
        // gram_finish_kernel walks 1 024 partials one load at a time)
        hipLaunchKernelGGL(weighted_colsum_finish_kernel, grid1((uint64_t)n, 64), dim3(256), 0, st.stream, slab, blocks, 1u, n, C, n);
        SCANRS_HIP(hipGetLastError());
        return;
    }
    // work queued on a side stream runs beside the persistent tile kernel of a sparse pass, which holds all of every CU's LDS: a
    // kernel that needs LDS waits for the end of the pass; the register-only MFMA kernels below fit the registers the tile kernel leaves
    const bool side = st.dense_side_no_lds && ((st.aux_stream && st.stream == st.aux_stream) || (st.aux2_stream && st.stream == st.aux2_stream));
    if (!side && gram_tiled_ok(n, m, rows) && !(ldx & 1u) && !(ldy & 1u)) {
        launch_gram_tiled(st, X, ldx, n, Y, ldy, m, rows, C);
        return;
    }
    const uint32_t tiles_n = (n + 31u) / 32u, tiles_m = (m + 31u) / 32u;
    const uint64_t tiles = (uint64_t)tiles_n * tiles_m;
    // aim for ~8k waves, at least 64 rows per slice
    uint64_t splits = std::max<uint64_t>(1, std::min<uint64_t>((rows + 63) / 64, (8192 + tiles - 1) / tiles));
    splits = std::min<uint64_t>(splits, 1024);
    uint64_t rps = (rows + splits - 1) / splits;
    rps = (rps + 7) & ~7ull;
    if (rps == 0) rps = 8;
    splits = std::max<uint64_t>(1, (rows + rps - 1) / rps);
    double *slab = st.scratch.get<double>(st.skey("gram_slab"), (size_t)splits * n * m);
    {
        ProfScope ps(st, "gram_mfma_f64", (double)rows * (n + (X == Y && n == m ? 0 : m)) * 8.0 + (double)n * m * 8.0);
        hipLaunchKernelGGL(gram_kernel, dim3((unsigned)tiles, (unsigned)splits), dim3(64), 0, st.stream, X, ldx, n, Y, ldy,
                           m, rows, rps, tiles_m, slab, st.skip_flag);
        hipLaunchKernelGGL(gram_finish_kernel, grid1((uint64_t)n * m, 256), dim3(256), 0, st.stream, slab,
                           (uint32_t)splits, (uint64_t)n * m, C, st.skip_flag);
    }
    SCANRS_HIP(hipGetLastError());
}

void launch_gemm_nn(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *W, uint32_t ldw, uint32_t m,
                    uint64_t rows, double alpha, double beta, const double *Cin, uint32_t ldc, double *Out,
                    uint32_t ldo) {
    if (rows == 0 || m == 0) return;
    if (ldx & 1u) fail(SCANRS_ERR_ARGUMENT, "gemm: ldx must be even");
    if (st.gemm_direct && gemm_direct_ok(X, ldx, n, m, rows)) {
        launch_gemm_direct(st, X, ldx, n, W, ldw, m, rows, alpha, beta, Cin, ldc, Out, ldo);
        return;
    }
    const bool side = st.dense_side_no_lds && ((st.aux_stream && st.stream == st.aux_stream) || (st.aux2_stream && st.stream == st.aux2_stream));
    if (!side && gemm_tiled_ok(n, m, rows)) {
        launch_gemm_tiled(st, X, ldx, n, W, ldw, m, rows, alpha, beta, Cin, ldc, Out, ldo);
        return;
    }
    const uint64_t waves = (rows + 15) / 16;
    ProfScope ps(st, "gemm_nn_mfma_f64", (double)rows * (n + m) * 8.0 + (double)n * m * 8.0);
    const dim3 block(256);
    if (m <= 16) {
        hipLaunchKernelGGL((gemm_nn_kernel<1>), dim3((unsigned)((waves + 3) / 4), (m + 15u) / 16u), block, 0, st.stream, X,
                           ldx, n, W, ldw, m, rows, alpha, beta, Cin, ldc, Out, ldo, st.skip_flag);
    } else if (m <= 32) {
        hipLaunchKernelGGL((gemm_nn_kernel<2>), dim3((unsigned)((waves + 3) / 4), (m + 31u) / 32u), block, 0, st.stream, X,
                           ldx, n, W, ldw, m, rows, alpha, beta, Cin, ldc, Out, ldo, st.skip_flag);
    } else {
        hipLaunchKernelGGL((gemm_nn_kernel<4>), dim3((unsigned)((waves + 3) / 4), (m + 63u) / 64u), block, 0, st.stream, X,
                           ldx, n, W, ldw, m, rows, alpha, beta, Cin, ldc, Out, ldo, st.skip_flag);
    }
    SCANRS_HIP(hipGetLastError());
}

// one column with a scalar that arrives as a kernel argument (IRLBA's normalisations and three-term updates, irlba.rs:137-160: the
// scalar used to travel as a 1 x 1 matrix through a pageable upload and a synchronisation per call):
//   MODE 0: dst[r] = src[r] * alpha        MODE 1: dst[r] = src[r] * alpha + dst[r]  (product rounded, then the sum: what the GEMM form gave)
template <int MODE>
__global__ void col_scalar_kernel(double *dst, uint32_t ldd, const double *src, uint32_t lds, uint64_t rows, double alpha) { // (dst may be src)
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const double t = __dmul_rn(src[r * lds], alpha);
    dst[r * ldd] = MODE == 0 ? t : __dadd_rn(t, dst[r * ldd]);
}
// the same with the scalar taken from device memory as the SQUARE of a norm (a 1 x 1 Gram matrix left where the Gram kernel wrote it):
//   MODE 0: dst[r] = src[r] * invcheck(sqrt(*nsq))  (irlba.rs:25-33: 1 / x above 2 eps, else 0)    MODE 1: dst[r] = src[r] * (-sqrt(*nsq)) + dst[r]
template <int MODE>
__global__ void col_scalar_dev_kernel(double *dst, uint32_t ldd, const double *src, uint32_t lds, uint64_t rows, const double *__restrict__ nsq) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const double nrm = __dsqrt_rn(*nsq);
    const double alpha = MODE == 0 ? (nrm > 2.0 * 2.220446049250313e-16 ? __ddiv_rn(1.0, nrm) : 0.0) : -nrm;
    const double t = __dmul_rn(src[r * lds], alpha);
    dst[r * ldd] = MODE == 0 ? t : __dadd_rn(t, dst[r * ldd]);
}
void launch_col_scale_dev(Storage &st, double *dst, uint32_t ldd, const double *src, uint32_t lds, uint64_t rows, const double *nsq) {
    if (rows == 0) return;
    hipLaunchKernelGGL(col_scalar_dev_kernel<0>, grid1(rows, 256), dim3(256), 0, st.stream, dst, ldd, src, lds, rows, nsq);
    SCANRS_HIP(hipGetLastError());
}
void launch_col_axpy_dev(Storage &st, double *y, uint32_t ldy, const double *x, uint32_t ldx, uint64_t rows, const double *nsq) {
    if (rows == 0) return;
    hipLaunchKernelGGL(col_scalar_dev_kernel<1>, grid1(rows, 256), dim3(256), 0, st.stream, y, ldy, x, ldx, rows, nsq);
    SCANRS_HIP(hipGetLastError());
}
void launch_col_scale(Storage &st, double *dst, uint32_t ldd, const double *src, uint32_t lds, uint64_t rows, double alpha) {
    if (rows == 0) return;
    hipLaunchKernelGGL(col_scalar_kernel<0>, grid1(rows, 256), dim3(256), 0, st.stream, dst, ldd, src, lds, rows, alpha);
    SCANRS_HIP(hipGetLastError());
}
void launch_col_axpy(Storage &st, double *y, uint32_t ldy, const double *x, uint32_t ldx, uint64_t rows, double alpha) {
    if (rows == 0) return;
    hipLaunchKernelGGL(col_scalar_kernel<1>, grid1(rows, 256), dim3(256), 0, st.stream, y, ldy, x, ldx, rows, alpha);
    SCANRS_HIP(hipGetLastError());
}
void launch_copy_cols(Storage &st, const double *src, uint32_t lds, double *dst, uint32_t ldd, uint64_t rows,
                      uint32_t l) {
    if (rows == 0 || l == 0) return;
    hipLaunchKernelGGL(copy_cols_kernel, grid1(rows * l, 256), dim3(256), 0, st.stream, src, lds, dst, ldd, rows, l, st.skip_flag);
    SCANRS_HIP(hipGetLastError());
}
void launch_permute_cols(Storage &st, const double *src, uint32_t lds, double *dst, uint32_t ldd, uint64_t rows,
                         const uint32_t *d_idx, uint32_t n_idx, bool scatter) {
    if (rows == 0 || n_idx == 0) return;
    hipLaunchKernelGGL(permute_cols_kernel, grid1(rows * n_idx, 256), dim3(256), 0, st.stream, src, lds, dst, ldd, rows, d_idx,
                       n_idx, scatter ? 1 : 0);
    SCANRS_HIP(hipGetLastError());
}
void launch_omega_jump(Storage &st, const uint64_t *d_jpow, int n_pow, const uint64_t s[4], uint64_t d, uint64_t total, double *out,
                       uint32_t ld, uint64_t seq_cols, bool transpose) {
    const uint64_t streams = (total + d - 1) / d;
    hipLaunchKernelGGL(omega_jump_kernel, grid1(streams, 256), dim3(256), 0, st.stream, d_jpow, n_pow, s[0], s[1], s[2], s[3], d, total,
                       out, ld, seq_cols, transpose ? 1 : 0);
    SCANRS_HIP(hipGetLastError());
}
void launch_transpose(Storage &st, const double *src, uint64_t rows, uint64_t cols, double *dst, uint32_t ldd) {
    if (rows == 0 || cols == 0) return;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(256), 0, st.stream, src,
                       rows, cols, dst, ldd);
    SCANRS_HIP(hipGetLastError());
}
// columns [c0, c0 + nc) of a panel filled with a fixed pseudo-random function of (GLOBAL row, column): every rank of a sharded
// panel produces its own rows of the same matrix without talking to anybody (splitmix64 of the pair, mapped to (-1, 1))
__global__ void fill_hash_kernel(double *p, uint32_t ld, uint64_t rows, uint64_t row0, uint32_t c0, uint32_t nc, uint64_t seed) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * nc) return;
    const uint64_t r = e / nc;
    const uint32_t c = (uint32_t)(e - r * nc);
    uint64_t z = seed + (row0 + r) * 0x9E3779B97F4A7C15ull + (uint64_t)(c0 + c) * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    p[r * ld + c0 + c] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
void launch_fill_hash(Storage &st, double *p, uint32_t ld, uint64_t rows, uint64_t row0, uint32_t c0, uint32_t nc, uint64_t seed) {
    if (rows == 0 || nc == 0) return;
    hipLaunchKernelGGL(fill_hash_kernel, grid1(rows * nc, 256), dim3(256), 0, st.stream, p, ld, rows, row0, c0, nc, seed);
}
void launch_fill_f64(Storage &st, double *p, uint64_t n, double v) {
    if (n == 0) return;
    hipLaunchKernelGGL(fill_kernel, grid1(n, 256), dim3(256), 0, st.stream, p, n, v);
    SCANRS_HIP(hipGetLastError());
}
// ---- per-INNER-position sums of mapped values from the copy whose outer vectors are the summed-over axis ----------------------
// The moments of normalize() (sum and sum of squares of log(1 + x s_c) per gene over the cells) walk the gene-major copy and
// pay one f64 logarithm per nonzero (5.9 ms at 10^9 nonzeros: the pass is bound by that arithmetic, not by its 8 B per
// nonzero). On the CELL-major copy the logarithm's argument depends on the count and the outer vector only, and 99.9 % of
// the counts are at most 8: a wave evaluates a cell's eight values once and every nonzero is a lookup — but the sums belong
// to the inner positions, i.e. they are scattered. They are collected in LDS (a workgroup owns a range of 8192 inner positions
// and a block of cells; the nonzeros of a cell inside the range are a contiguous run: bounds table) as 64-bit FIXED-POINT
// numbers with integer atomics: integer addition is associative, so the result does not depend on the order in which the waves
// arrive (float atomics would make it depend on timing), and with the scale chosen per launch from a bound of the values
// (col_moments_plan) a workgroup's sums cannot overflow and round each term below 2^-30 of the largest value; the partial sums
// of the workgroups are added in double, in workgroup order.
constexpr uint32_t CM_RANGE_SHIFT = 13; // inner positions per range: 8192 x 2 x 8 B = 128 KB of LDS
constexpr uint32_t CM_TAB = 8;          // counts 1 .. 8 come from the per-cell table
constexpr uint32_t CM_NV = 4;           // cells per wave and trip (their bounds, their segments' loads side by side)
template <int MODE>
__global__ __launch_bounds__(1024) void col_moments_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                           const uint32_t *__restrict__ values, const uint32_t *__restrict__ bounds, uint32_t nb,
                                                           uint64_t n_outer, uint64_t cells_per_wg, uint32_t n_inner, DevMap map, double scale1,
                                                           double scale2, unsigned long long *__restrict__ slab) {
    extern __shared__ unsigned long long cm_acc[]; // [MODE][inner position of the range]: one array per moment (interleaved, the 16-byte stride put 64 lanes on 16 bank groups)
    const uint32_t range = blockIdx.y, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const uint32_t g0 = range << CM_RANGE_SHIFT, ng = min(1u << CM_RANGE_SHIFT, n_inner - g0);
    const uint32_t tiles_per_range = (1u << CM_RANGE_SHIFT) >> BT_SHIFT;
    const uint32_t b0 = range * tiles_per_range, b1 = min(nb, b0 + tiles_per_range);
    for (uint32_t i = tid; i < (MODE << CM_RANGE_SHIFT); i += blockDim.x) cm_acc[i] = 0ull;
    __syncthreads();
    const uint64_t c_begin = (uint64_t)wg * cells_per_wg, c_end = min(n_outer, c_begin + cells_per_wg);
    for (uint64_t c4 = c_begin + (uint64_t)wave * CM_NV; c4 < c_end; c4 += (uint64_t)n_waves * CM_NV) {
        uint32_t len_l = 0, blo = 0, bhi = 0;
        if (lane < CM_NV && c4 + lane < c_end) {
            const uint32_t *bd = bounds + (c4 + lane) * (nb + 1);
            const uint32_t o0 = bd[b0];
            len_l = bd[b1] - o0;
            const uint64_t base = indptr[c4 + lane] + o0;
            blo = (uint32_t)base;
            bhi = (uint32_t)(base >> 32);
        }
        // lanes 8 r .. 8 r + 7: the chain at counts 1 .. 8 of cell r of this trip
        double tab = 0.0;
        if (lane < CM_NV * CM_TAB && c4 + lane / CM_TAB < c_end) tab = eval_map(map, (lane % CM_TAB) + 1u, (uint32_t)(c4 + lane / CM_TAB), 0u);
        uint32_t start[CM_NV + 1];
        uint64_t base[CM_NV];
        start[0] = 0;
#pragma unroll
        for (uint32_t r = 0; r < CM_NV; r++) {
            start[r + 1] = start[r] + rdlane(len_l, r);
            base[r] = ((uint64_t)rdlane(bhi, r) << 32) | rdlane(blo, r);
        }
        const uint32_t total = start[CM_NV];
        // four strides of 64 nonzeros per trip: their index and count loads go out together (one stride per trip made the walk a
        // chain of load latencies: 3.9 ms per pass instead of 2.x)
        constexpr uint32_t CM_U = 8;
        for (uint32_t q0 = 0; q0 < total; q0 += 64u * CM_U) {
            uint32_t g[CM_U], cnt[CM_U], r[CM_U];
            bool on[CM_U];
#pragma unroll
            for (uint32_t u = 0; u < CM_U; u++) {
                const uint32_t p_raw = q0 + 64u * u + lane;
                on[u] = p_raw < total;
                // no branch around the loads (a lane past the end re-reads the trip's last nonzero and is masked below): behind a
                // branch every stride waited for its own index before the next stride's loads went out
                const uint32_t p = on[u] ? p_raw : total - 1u;
                r[u] = 0;
                uint64_t a = base[0] + p;
#pragma unroll
                for (uint32_t rr = 1; rr < CM_NV; rr++)
                    if (p >= start[rr]) {
                        r[u] = rr;
                        a = base[rr] + (p - start[rr]);
                    }
                g[u] = indices[a] - g0;
                cnt[u] = values[a];
            }
#pragma unroll
            for (uint32_t u = 0; u < CM_U; u++) {
                if (q0 + 64u * u >= total) break; // uniform
                double v = __shfl(tab, (int)(r[u] * CM_TAB + min(cnt[u], CM_TAB) - 1u)); // every lane takes part in the exchange
                if (on[u]) {
                    if (cnt[u] > CM_TAB) v = eval_map(map, cnt[u], (uint32_t)(c4 + r[u]), 0u);
                    // 0 <= t < 2^51 (col_moments_plan): t + 2^52 holds round-to-nearest(t) in its mantissa — two instructions
                    // instead of the ~15 of a general f64 -> i64 conversion, which made this loop VALU-bound
                    atomicAdd(&cm_acc[g[u]], (unsigned long long)__double_as_longlong(v * scale1 + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull);
                    if (MODE == 2)
                        atomicAdd(&cm_acc[(1u << CM_RANGE_SHIFT) + g[u]],
                                  (unsigned long long)__double_as_longlong(v * v * scale2 + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull);
                }
            }
        }
    }
    __syncthreads();
    unsigned long long *dst = slab + ((size_t)range * gridDim.x + wg) * ((size_t)MODE << CM_RANGE_SHIFT);
    for (uint32_t i = tid; i < (MODE << CM_RANGE_SHIFT); i += blockDim.x) dst[i] = cm_acc[i];
    (void)ng;
}
// the partial sums of the workgroups, added in double in workgroup order
template <int MODE>
__global__ void col_moments_finish_kernel(const unsigned long long *__restrict__ slab, uint32_t n_wg, uint32_t n_inner, double inv1, double inv2,
                                          double *__restrict__ out_sum, double *__restrict__ out_sumsq) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_inner) return;
    const uint32_t range = g >> CM_RANGE_SHIFT, gi = g & ((1u << CM_RANGE_SHIFT) - 1u);
    double s1 = 0.0, s2 = 0.0;
    for (uint32_t w = 0; w < n_wg; w++) {
        const unsigned long long *src = slab + ((size_t)range * n_wg + w) * ((size_t)MODE << CM_RANGE_SHIFT) + gi;
        s1 += (double)(long long)src[0];
        if (MODE == 2) s2 += (double)(long long)src[1u << CM_RANGE_SHIFT];
    }
    out_sum[g] = s1 * inv1;
    if (MODE == 2 && out_sumsq) out_sumsq[g] = s2 * inv2;
}
__global__ void max_u32_kernel(const uint32_t *__restrict__ v, uint64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) m = max(m, v[i]);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63u) == 0) atomicMax(out, m);
}
// [0] = smallest, [1] = largest value as order-preserving bit patterns of non-negative doubles; [2] != 0: a negative or non-finite value
__global__ void minmax_nonneg_kernel(const double *__restrict__ v, uint64_t n, unsigned long long *__restrict__ out) {
    unsigned long long lo = ~0ull, hi = 0ull;
    int bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double x = v[i];
        if (!(x >= 0.0) || !isfinite(x)) {
            bad = 1;
        } else {
            const unsigned long long b = (unsigned long long)__double_as_longlong(x);
            lo = b < lo ? b : lo;
            hi = b > hi ? b : hi;
        }
    }
    atomicMin(&out[0], lo);
    atomicMax(&out[1], hi);
    if (bad) atomicMax(&out[2], 1ull);
}

// Can the per-inner sums of `map` over `cp` (outer = the summed-over axis) take col_moments_kernel? Every link must be a
// ScaleAxis over the OUTER position with non-negative finite factors or a logarithm (then the value is non-negative and grows
// with the count and with the factors: its largest possible value follows from the largest count and the largest factors).
// Returns the fixed-point scales in s1 / s2; false = use the ordinary pass.
static bool col_moments_plan(Storage &st, SparseCopy &cp, const DevMap &map, int mode, uint32_t n_wg, double &s1, double &s2) {
    bool has_log = false;
    for (int i = 0; i < map.n; i++) {
        const int k = map.ops[i].kind;
        if (k == OP_SCALE_AXIS) {
            if (!map.ops[i].a_outer) return false;
        } else if (k == OP_LN_1P || k == OP_LOG2_1P || k == OP_LOG10_1P) {
            has_log = true;
        } else {
            return false;
        }
    }
    if (!has_log || cp.n_outer == 0 || cp.n_inner == 0) return false;
    if (cp.max_value == 0) { // structure only: once per copy
        uint32_t *d = st.scratch.get<uint32_t>("cm_maxv", 1);
        SCANRS_HIP(hipMemsetAsync(d, 0, 4, st.stream));
        hipLaunchKernelGGL(max_u32_kernel, dim3(2048), dim3(256), 0, st.stream, cp.values.p, cp.nnz, d);
        const uint32_t h = SCANRS_D2H_VALUE(d, st.stream);
        cp.max_value = std::max(h, 1u);
    }
    double x = (double)cp.max_value;
    unsigned long long *d = st.scratch.get<unsigned long long>("cm_minmax", 3);
    for (int i = 0; i < map.n; i++) {
        const int k = map.ops[i].kind;
        if (k == OP_SCALE_AXIS) {
            const unsigned long long init[3] = {~0ull, 0ull, 0ull};
            SCANRS_HIP(hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, st.stream));
            hipLaunchKernelGGL(minmax_nonneg_kernel, dim3(512), dim3(256), 0, st.stream, map.ops[i].a, cp.n_outer, d);
            struct H3 {
                unsigned long long v[3];
            };
            const H3 h3 = SCANRS_D2H_VALUE(reinterpret_cast<const H3 *>(d), st.stream);
            const unsigned long long *h = h3.v;
            if (h[2]) return false;
            double amax;
            memcpy(&amax, &h[1], 8);
            x *= amax;
        } else if (k == OP_LN_1P) {
            x = std::log1p(x);
        } else if (k == OP_LOG2_1P) {
            x = std::log2(1.0 + x);
        } else {
            x = std::log10(1.0 + x);
        }
        if (!std::isfinite(x)) return false;
    }
    // a workgroup adds at most cells_per_wg values of at most x (x^2): keep the sums below 2^62 and the terms' rounding at 2^-30 of x or finer
    const double cells = std::ceil((double)cp.n_outer / n_wg);
    const double b1 = std::max(x, 1e-300) * cells * 1.0001, b2 = std::max(x * x, 1e-300) * cells * 1.0001;
    // ... and every term below 2^51 (the kernel's conversion)
    const int e1 = std::min(62 - (int)std::ceil(std::log2(b1)), 50 - (int)std::ceil(std::log2(std::max(x, 1e-300) * 1.0001)));
    const int e2 = std::min(62 - (int)std::ceil(std::log2(b2)), 50 - (int)std::ceil(std::log2(std::max(x * x, 1e-300) * 1.0001)));
    if (e1 > 1000 || e2 > 1000 || (double)e1 + std::log2(std::max(x, 1e-300)) < 30.0 || (mode == 2 && (double)e2 + std::log2(std::max(x * x, 1e-300)) < 30.0)) return false;
    s1 = std::ldexp(1.0, e1);
    s2 = std::ldexp(1.0, e2);
    return true;
}
static uint32_t ensure_bounds(Storage &st, SparseCopy &cp, hipStream_t s);
// out_sum[i] (and out_sumsq[i]) over the outer vectors of cp, per INNER position i. false: not eligible (nothing was launched).
bool launch_col_moments(Storage &st, SparseCopy &cp, const DevMap &map, int mode, double *out_sum, double *out_sumsq) {
    if (mode != 1 && mode != 2) return false;
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t n_ranges = (uint32_t)((cp.n_inner + (1ull << CM_RANGE_SHIFT) - 1) >> CM_RANGE_SHIFT);
    if (cp.n_inner >= (1ull << 31) || n_ranges > 65535u) return false;
    // one workgroup per CU and range at a time; the cells dealt over as many workgroups as keep every CU busy for all ranges
    const uint32_t n_wg = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_cu, (cp.n_outer + 63) / 64));
    double s1 = 0.0, s2 = 0.0;
    if (!col_moments_plan(st, cp, map, mode, n_wg, s1, s2)) return false;
    const uint32_t nb = ensure_bounds(st, cp);
    const uint64_t cells_per_wg = (cp.n_outer + n_wg - 1) / n_wg;
    unsigned long long *slab = st.scratch.get<unsigned long long>("cm_slab", (size_t)n_ranges * n_wg * ((size_t)mode << CM_RANGE_SHIFT));
    const size_t shmem = ((size_t)mode << CM_RANGE_SHIFT) * 8;
    ProfScope ps(st, mode == 2 ? "col_moments" : "col_sums", (double)cp.nnz * 8.0 + (double)cp.n_outer * (n_ranges + 1) * 4.0);
    if (mode == 2) {
        SCANRS_HIP(hipFuncSetAttribute((const void *)col_moments_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((col_moments_kernel<2>), dim3(n_wg, n_ranges), dim3(1024), shmem, st.stream, cp.indptr.p, cp.indices.p, cp.values.p, cp.bounds.p, nb,
                           cp.n_outer, cells_per_wg, (uint32_t)cp.n_inner, map, s1, s2, slab);
        hipLaunchKernelGGL((col_moments_finish_kernel<2>), dim3((unsigned)((cp.n_inner + 255) / 256)), dim3(256), 0, st.stream, slab, n_wg, (uint32_t)cp.n_inner,
                           1.0 / s1, 1.0 / s2, out_sum, out_sumsq);
    } else {
        SCANRS_HIP(hipFuncSetAttribute((const void *)col_moments_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((col_moments_kernel<1>), dim3(n_wg, n_ranges), dim3(1024), shmem, st.stream, cp.indptr.p, cp.indices.p, cp.values.p, cp.bounds.p, nb,
                           cp.n_outer, cells_per_wg, (uint32_t)cp.n_inner, map, s1, s2, slab);
        hipLaunchKernelGGL((col_moments_finish_kernel<1>), dim3((unsigned)((cp.n_inner + 255) / 256)), dim3(256), 0, st.stream, slab, n_wg, (uint32_t)cp.n_inner,
                           1.0 / s1, 1.0 / s2, out_sum, (double *)nullptr);
    }
    SCANRS_HIP(hipGetLastError());
    return true;
}

void launch_finish_moments(Storage &st, const double *sum, const double *sumsq, uint64_t n, double m, int given_scale,
                           const double *scale_in, double *neg_mean_over_scale, double *inv_scale, double *scale_out) {
    if (n == 0) return;
    hipLaunchKernelGGL(finish_moments_kernel, grid1(n, 256), dim3(256), 0, st.stream, sum, sumsq, n, m, given_scale,
                       scale_in, neg_mean_over_scale, inv_scale, scale_out);
    SCANRS_HIP(hipGetLastError());
}
void launch_u32_to_scale(Storage &st, const uint32_t *counts, uint64_t n, double target, double *out) {
    if (n == 0) return;
    hipLaunchKernelGGL(u32_to_scale_kernel, grid1(n, 256), dim3(256), 0, st.stream, counts, n, target, out);
    SCANRS_HIP(hipGetLastError());
}
void launch_hist12(Storage &st, const uint32_t *v, uint64_t n, uint32_t shift, uint32_t digit_mask, uint32_t prefix_mask,
                   uint32_t prefix, unsigned long long *hist) {
    SCANRS_HIP(hipMemsetAsync(hist, 0, 4096 * sizeof(unsigned long long), st.stream));
    if (n) {
        const unsigned blocks = (unsigned)std::min<uint64_t>(1024, (n + 255) / 256);
        hipLaunchKernelGGL(hist12_kernel, dim3(blocks), dim3(256), 0, st.stream, v, n, shift, digit_mask, prefix_mask, prefix, hist);
    }
    SCANRS_HIP(hipGetLastError());
}
void launch_densify(Storage &st, const SparseCopy &cp, const DevMap &map, bool outer_is_view_row, uint64_t cols_v,
                    double *out) {
    if (cp.n_items == 0) return;
    hipLaunchKernelGGL(densify_kernel, dim3((cp.n_items + 3u) / 4u), dim3(256), 0, st.stream, cp.indices.p, cp.values.p,
                       cp.items.p, cp.n_items, map, outer_is_view_row ? 1 : 0, cols_v, out);
    SCANRS_HIP(hipGetLastError());
}
void launch_binom_uv(Storage &st, int kind, const double *n, uint64_t ncols, const double *rowsum, uint64_t nrows,
                     double total, double *pi, double *u, double *v) {
    if (nrows) hipLaunchKernelGGL(binom_rows_kernel, grid1(nrows, 256), dim3(256), 0, st.stream, kind, rowsum, nrows, total, pi, u);
    if (ncols) hipLaunchKernelGGL(binom_cols_kernel, grid1(ncols, 256), dim3(256), 0, st.stream, kind, n, ncols, v);
    SCANRS_HIP(hipGetLastError());
}
void launch_sum_f64(Storage &st, const double *x, uint64_t n, double *out) {
    hipLaunchKernelGGL(sum_f64_kernel, dim3(1), dim3(256), 0, st.stream, x, n, out);
    SCANRS_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// storage management
// One-shot all-reduce of the single-process group (comm.cpp): rank r sums slice r of every rank's buffer in rank order
// and writes the sum back into every buffer. Peer buffers are reached through peer-mapped pointers (xGMI) or are on the
// same device; the caller brackets the launch with group barriers.
struct PeerTable {
    void *p[16];
};
template <typename T>
__global__ __launch_bounds__(256) void local_allreduce_kernel(PeerTable tab, uint32_t world, uint64_t lo, uint64_t hi) {
    for (uint64_t e = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < hi; e += (uint64_t)gridDim.x * blockDim.x) {
        T acc = static_cast<const T *>(tab.p[0])[e];
        for (uint32_t r = 1; r < world; r++) acc += static_cast<const T *>(tab.p[r])[e];
        for (uint32_t r = 0; r < world; r++) static_cast<T *>(tab.p[r])[e] = acc;
    }
}
void launch_local_allreduce(hipStream_t s, void *const *bufs, uint32_t world, uint32_t rank, uint64_t count, int dtype) {
    if (world > 16) fail(SCANRS_ERR_ARGUMENT, "single-process groups hold at most 16 shards");
    PeerTable tab;
    for (uint32_t r = 0; r < 16; r++) tab.p[r] = r < world ? bufs[r] : nullptr;
    const uint64_t lo = count * rank / world, hi = count * (rank + 1) / world;
    if (hi <= lo) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>(1024, (hi - lo + 255) / 256);
    if (dtype == 0)
        hipLaunchKernelGGL(local_allreduce_kernel<double>, dim3(blocks), dim3(256), 0, s, tab, world, lo, hi);
    else
        hipLaunchKernelGGL(local_allreduce_kernel<unsigned long long>, dim3(blocks), dim3(256), 0, s, tab, world, lo, hi);
    SCANRS_HIP(hipGetLastError());
}

// Work items of the gather kernels, built on the device: per outer vector the number of pieces (1, or ceil(len / ITEM_NNZ) for a
// long vector, which also gets a MultiRow and slab rows), two exclusive scans, one fill pass — the host only reads the three
// totals (it used to walk indptr on one core: 8 MB down, a loop over every vector, 24 MB of items up, ~15 ms at 10^6 cells).
__global__ void item_count_kernel(const uint64_t *__restrict__ indptr, uint64_t n_outer, unsigned long long *__restrict__ pieces,
                                  unsigned long long *__restrict__ multi_slab) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o > n_outer) return;
    if (o == n_outer) { // the scans run over n_outer + 1 entries: the last one receives the totals
        pieces[o] = 0;
        multi_slab[o] = 0;
        return;
    }
    const uint64_t len = indptr[o + 1] - indptr[o];
    const bool multi = len > ITEM_NNZ;
    const unsigned long long np = multi ? (len + ITEM_NNZ - 1) / ITEM_NNZ : 1ull;
    pieces[o] = np;
    multi_slab[o] = multi ? (1ull | (np << 32)) : 0ull; // low word: MultiRows so far, high word: slab rows so far
}
__global__ void item_fill_kernel(const uint64_t *__restrict__ indptr, uint64_t n_outer, const unsigned long long *__restrict__ item_off,
                                 const unsigned long long *__restrict__ multi_slab_off, Item *__restrict__ items, MultiRow *__restrict__ multi) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_outer) return;
    const uint64_t a = indptr[o], b = indptr[o + 1], len = b - a;
    const uint64_t at = item_off[o];
    if (len <= ITEM_NNZ) {
        items[at] = Item{a, (uint32_t)o, (uint32_t)len, NO_SLAB, 0};
        return;
    }
    const uint32_t pieces = (uint32_t)((len + ITEM_NNZ - 1) / ITEM_NNZ);
    const unsigned long long ms = multi_slab_off[o];
    const uint32_t slab0 = (uint32_t)(ms >> 32);
    multi[(uint32_t)ms] = MultiRow{(uint32_t)o, slab0, pieces, 0};
    for (uint32_t k = 0; k < pieces; k++) {
        const uint64_t s0 = a + (uint64_t)k * ITEM_NNZ;
        const uint64_t ln = b - s0 < ITEM_NNZ ? b - s0 : ITEM_NNZ;
        items[at + k] = Item{s0, (uint32_t)o, (uint32_t)ln, slab0 + k, 0};
    }
}

void SparseCopy::build_items(hipStream_t s) {
    n_items = n_multi = n_slab = 0;
    if (n_outer == 0) {
        items.alloc(0);
        multi.alloc(0);
        return;
    }
    const size_t n = (size_t)n_outer + 1;
    DevBuf<unsigned long long> cnt(2 * n), off(2 * n);
    const dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(item_count_kernel, grid, dim3(256), 0, s, indptr.p, n_outer, cnt.p, cnt.p + n);
    size_t tmp_bytes = 0;
    SCANRS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, cnt.p, off.p, 0ull, n, rocprim::plus<unsigned long long>(), s));
    DevBuf<char> tmp(std::max<size_t>(tmp_bytes, 16));
    SCANRS_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, cnt.p, off.p, 0ull, n, rocprim::plus<unsigned long long>(), s));
    SCANRS_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, cnt.p + n, off.p + n, 0ull, n, rocprim::plus<unsigned long long>(), s));
    const unsigned long long tot[2] = {SCANRS_D2H_VALUE(off.p + n_outer, s), SCANRS_D2H_VALUE(off.p + n + n_outer, s)};
    if (tot[0] > 0xFFFFFFFFull) fail(SCANRS_ERR_SHAPE, "matrix needs more than 2^32 - 1 work items");
    n_items = (uint32_t)tot[0];
    n_multi = (uint32_t)tot[1];
    n_slab = (uint32_t)(tot[1] >> 32);
    items.alloc(n_items);
    multi.alloc(n_multi);
    hipLaunchKernelGGL(item_fill_kernel, grid, dim3(256), 0, s, indptr.p, n_outer, off.p, off.p + n, items.p, multi.p);
    SCANRS_HIP(hipGetLastError());
    SCANRS_SYNC(s); // the temporaries are released on return
}

void validate_copy(Storage &st, SparseCopy &cp, uint64_t *zeros, uint64_t *bad) {
    DevBuf<unsigned long long> c(4);
    SCANRS_HIP(hipMemsetAsync(c.p, 0, 4 * sizeof(unsigned long long), st.stream));
    if (cp.n_outer) {
        hipLaunchKernelGGL(validate_starts_kernel, grid1(cp.n_outer, 256), dim3(256), 0, st.stream, cp.indptr.p, cp.n_outer, cp.indices.p, cp.nnz, c.p);
        // a broken indptr first: the streaming pass below trusts nnz only, the later passes trust indptr
        const unsigned long long h1 = SCANRS_D2H_VALUE(c.p + 1, st.stream);
        if (h1) {
            *zeros = 0;
            *bad = h1;
            return;
        }
    }
    if (cp.nnz) {
        const uint64_t n4 = (cp.nnz + 3) / 4;
        int dev = 0, n_cu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        const unsigned blocks = (unsigned)std::min<uint64_t>((n4 + 255) / 256, (uint64_t)n_cu * 16u);
        hipLaunchKernelGGL(validate_stream_kernel, dim3(blocks), dim3(256), 0, st.stream, cp.indices.p, cp.values.p, cp.nnz, cp.n_inner, c.p);
    }
    struct H4 {
        unsigned long long v[4];
    };
    const H4 h4 = SCANRS_D2H_VALUE(reinterpret_cast<const H4 *>(c.p), st.stream);
    const unsigned long long *h = h4.v;
    *zeros = h[0];
    *bad = h[1] - std::min(h[1], h[2]);
    if (cp.nnz) cp.max_value = (uint32_t)std::max<unsigned long long>(1ull, h[3]);
}

void compact_nonzeros(Storage &st, SparseCopy &cp) {
    // Rare path (AbsIter skips stored zeros, sqz/src/vec.rs:113): rebuild on the host.
    std::vector<uint64_t> ip(cp.n_outer + 1);
    std::vector<uint32_t> ind(cp.nnz), val(cp.nnz);
    SCANRS_HIP(hipMemcpy(ip.data(), cp.indptr.p, ip.size() * 8, hipMemcpyDeviceToHost));
    if (cp.nnz) {
        SCANRS_HIP(hipMemcpy(ind.data(), cp.indices.p, cp.nnz * 4, hipMemcpyDeviceToHost));
        SCANRS_HIP(hipMemcpy(val.data(), cp.values.p, cp.nnz * 4, hipMemcpyDeviceToHost));
    }
    std::vector<uint64_t> np(cp.n_outer + 1, 0);
    uint64_t w = 0;
    for (uint64_t o = 0; o < cp.n_outer; o++) {
        np[o] = w;
        for (uint64_t p = ip[o]; p < ip[o + 1]; p++)
            if (val[p] != 0) {
                ind[w] = ind[p];
                val[w] = val[p];
                w++;
            }
    }
    np[cp.n_outer] = w;
    cp.nnz = w;
    SCANRS_HIP(hipMemcpy(cp.indptr.p, np.data(), np.size() * 8, hipMemcpyHostToDevice));
    if (w) {
        SCANRS_HIP(hipMemcpy(cp.indices.p, ind.data(), w * 4, hipMemcpyHostToDevice));
        SCANRS_HIP(hipMemcpy(cp.values.p, val.data(), w * 4, hipMemcpyHostToDevice));
    }
    (void)st;
}

// Sorts (index, count) pairs by index inside every outer vector: the repair the reference applies to 10x matrix files
// whose indices are unsorted within a cell (`new_from_unsorted_csc`, hdf5-io/src/matrix.rs:66-75).
void sort_outer_vectors(Storage &st, SparseCopy &cp) {
    if (cp.nnz == 0) return;
    if (cp.nnz > 0xFFFFFFFFull || cp.n_outer > 0xFFFFFFFFull) fail(SCANRS_ERR_SHAPE, "unsorted input beyond 2^32-1 nonzeros is not supported");
    DevBuf<uint32_t> keys_out, vals_out;
    keys_out.alloc(cp.nnz);
    vals_out.alloc(cp.nnz);
    uint32_t end_bit = 1;
    while (end_bit < 32u && (cp.n_inner >> end_bit) != 0) end_bit++;
    size_t tmp_bytes = 0;
    SCANRS_HIP(rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, cp.indices.p, keys_out.p, cp.values.p, vals_out.p, (unsigned)cp.nnz,
                                                   (unsigned)cp.n_outer, cp.indptr.p, cp.indptr.p + 1, 0u, end_bit, st.stream));
    DevBuf<unsigned char> tmp;
    tmp.alloc(tmp_bytes ? tmp_bytes : 1);
    SCANRS_HIP(rocprim::segmented_radix_sort_pairs(tmp.p, tmp_bytes, cp.indices.p, keys_out.p, cp.values.p, vals_out.p, (unsigned)cp.nnz,
                                                   (unsigned)cp.n_outer, cp.indptr.p, cp.indptr.p + 1, 0u, end_bit, st.stream));
    SCANRS_HIP(hipMemcpyAsync(cp.indices.p, keys_out.p, cp.nnz * 4, hipMemcpyDeviceToDevice, st.stream));
    SCANRS_HIP(hipMemcpyAsync(cp.values.p, vals_out.p, cp.nnz * 4, hipMemcpyDeviceToDevice, st.stream));
    SCANRS_SYNC(st.stream);
}

void build_transposed_copy(Storage &st, const SparseCopy &src, SparseCopy &dst, hipStream_t stream) {
    hipStream_t s = stream ? stream : st.stream;
    Tick tick("transposed copy build");
    dst.n_outer = src.n_inner;
    dst.n_inner = src.n_outer;
    dst.nnz = src.nnz;
    dst.indptr.alloc(dst.n_outer + 1);
    dst.indices.alloc(std::max<uint64_t>(1, dst.nnz));
    dst.values.alloc(std::max<uint64_t>(1, dst.nnz));
    const uint64_t nnz = src.nnz;
    if (nnz == 0) {
        SCANRS_HIP(hipMemsetAsync(dst.indptr.p, 0, (dst.n_outer + 1) * 8, s));
        dst.build_items(s);
        return;
    }
    auto bits_for = [](uint64_t maxv) { // bits that hold 0 .. maxv
        uint32_t b = 1;
        while (b < 64 && (maxv >> b) != 0) b++;
        return b;
    };
    const uint32_t ib = bits_for(src.n_inner ? src.n_inner - 1 : 0), ob = bits_for(src.n_outer ? src.n_outer - 1 : 0);
    const uint32_t cb = src.max_value ? bits_for(src.max_value) : 32u; // max_value is known for every copy that went through validation
    dst.max_value = src.max_value;
    if (ib + ob + cb <= 64u) {
        // one packed 64-bit key per nonzero, sorted on its inner-index bits (see pack_key_kernel)
        DevBuf<unsigned long long> ka(nnz), kb2(nnz);
        hipLaunchKernelGGL(pack_key_kernel, dim3((src.n_items + 3u) / 4u), dim3(256), 0, s, src.items.p, src.n_items, src.indices.p, src.values.p, ob, cb, ka.p);
        rocprim::double_buffer<unsigned long long> kb(ka.p, kb2.p);
        size_t tmp_bytes = 0;
        SCANRS_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, kb, (size_t)nnz, ob + cb, ob + cb + ib, s));
        DevBuf<char> tmp(std::max<size_t>(tmp_bytes, 16));
        SCANRS_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, kb, (size_t)nnz, ob + cb, ob + cb + ib, s));
        hipLaunchKernelGGL(unpack_key_kernel, grid1(nnz, 256), dim3(256), 0, s, kb.current(), nnz, ob, cb, dst.indices.p, dst.values.p);
        hipLaunchKernelGGL(lower_bound_key_kernel, grid1(dst.n_outer + 1, 256), dim3(256), 0, s, kb.current(), nnz, dst.n_outer, ob + cb, dst.indptr.p);
        SCANRS_HIP(hipGetLastError());
        SCANRS_SYNC(s);
    } else {
        // stable LSD radix sort of (inner index, (outer id, value)): inner vectors come out with
        // ascending outer ids, the order the reference's CSC kernel accumulates in (prod.rs:190-214).
        DevBuf<uint32_t> keys_a(nnz), keys_b(nnz);
        DevBuf<unsigned long long> vals_a(nnz), vals_b(nnz);
        SCANRS_HIP(hipMemcpyAsync(keys_a.p, src.indices.p, nnz * 4, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(pack_outer_kernel, dim3((src.n_items + 3u) / 4u), dim3(256), 0, s, src.items.p, src.n_items, src.values.p, vals_a.p);
        unsigned end_bit = 1;
        while (end_bit < 32 && (1ull << end_bit) < src.n_inner) end_bit++;
        rocprim::double_buffer<uint32_t> kb(keys_a.p, keys_b.p);
        rocprim::double_buffer<unsigned long long> vb(vals_a.p, vals_b.p);
        size_t tmp_bytes = 0;
        SCANRS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kb, vb, (size_t)nnz, 0u, end_bit, s));
        DevBuf<char> tmp(std::max<size_t>(tmp_bytes, 16));
        SCANRS_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, kb, vb, (size_t)nnz, 0u, end_bit, s));
        hipLaunchKernelGGL(unpack_kernel, grid1(nnz, 256), dim3(256), 0, s, vb.current(), nnz, dst.indices.p,
                           dst.values.p);
        hipLaunchKernelGGL(lower_bound_kernel, grid1(dst.n_outer + 1, 256), dim3(256), 0, s, kb.current(), nnz,
                           dst.n_outer, dst.indptr.p);
        SCANRS_HIP(hipGetLastError());
        SCANRS_SYNC(s);
    }
    dst.build_items(s);
}

// scanrs_init(): one empty launch per translation unit makes the runtime load this file's code object now instead of inside the
// first real call
__global__ void warm_kernels_kernel() {}
void warm_kernels(hipStream_t s) { hipLaunchKernelGGL(warm_kernels_kernel, dim3(1), dim3(64), 0, s); }

void warm_dense(hipStream_t s);
void warm_decode(hipStream_t s);
void warm_knn(hipStream_t s);
void warm_tiles(hipStream_t s);
void library_warm_up() {
    hipStream_t s = nullptr;
    SCANRS_HIP(hipStreamCreate(&s));
    warm_kernels(s);
    warm_dense(s);
    warm_decode(s);
    warm_knn(s);
    warm_tiles(s);
    const hipError_t e = hipGetLastError();
    SCANRS_SYNC(s);
    (void)hipStreamDestroy(s);
    if (e != hipSuccess) fail(SCANRS_ERR_DEVICE, "warm-up launch failed: %s", hipGetErrorString(e));
}

} // namespace scanrs
