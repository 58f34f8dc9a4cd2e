// LDS-tiled tall-skinny dense kernels on v_mfma_f64_16x16x4_f64 (gfx950): the big-panel versions of
// gram_kernel / gemm_nn_kernel in kernels.hip. They stand where the reference multiplies dense panels
// through ndarray `dot` -> BLAS (`T = Q^T A` Gram / `Q.dot(&U)`, scan-rs/src/dim_red/bk_svd.rs:134-142).
//
// Both compute C_tile(128 x 128) += sum_k A[k][i] * B[k][j] from two LDS images laid out [k][128 + pad]:
//   gram:  k = panel row,           A = X[:, i0:i0+128], B = Y[:, j0:j0+128]   (straight row copies)
//   gemm:  k = inner dimension n,   A[k][i] = X[r0+i][k0+k] (transposed while staging), B = W[k0+k][j0:]
// Workgroup = 4 waves, each owning a 64 x 64 quadrant as 4 x 4 MFMA tiles (64 f64 accumulators per lane).
// Operand maps of v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
// lane l holds D[row = (l>>4) + 4*reg][col = l&15]. LDS rows are padded to 144 doubles (1152 B) so that
// the two k values a 32-lane ds_read_b64 group touches fall on disjoint bank halves.
#include "common.hpp"

#include <algorithm>

namespace scanrs {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr uint32_t DT = 128;  // tile edge
constexpr uint32_t DK = 16;   // k-depth per staging step
constexpr uint32_t DLD = 144; // LDS row stride in doubles

__device__ __forceinline__ void tile_mma(const double *__restrict__ As, const double *__restrict__ Bs, uint32_t wi,
                                         uint32_t wj, uint32_t li, uint32_t lk, d4 (&acc)[4][4]) {
#pragma unroll
    for (uint32_t kk = 0; kk < DK / 4; kk++) {
        const uint32_t row = (kk * 4 + lk) * DLD;
        double a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            a[t] = As[row + wi * 64 + t * 16 + li];
            b[t] = Bs[row + wj * 64 + t * 16 + li];
        }
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int nj = 0; nj < 4; nj++) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
    }
}

// C = X^T Y over a slice of rows. grid.x = tile pair, grid.y = row split; partial tiles go to slab[split].
__global__ __launch_bounds__(256) void gram_tiled_kernel(const double *__restrict__ X, uint32_t ldx, uint32_t n,
                                                         const double *__restrict__ Y, uint32_t ldy, uint32_t m, uint64_t rows,
                                                         uint64_t rows_per_split, const uint32_t *__restrict__ tile_ij,
                                                         double *__restrict__ slab, const int *__restrict__ skip) {
    if (skip && *skip) return; // a converged orthonormalisation: its queued passes do nothing (Storage::skip_flag)
    __shared__ double As[DK * DLD];
    __shared__ double Bs[DK * DLD];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t li = lane & 15u, lk = lane >> 4, wi = w >> 1, wj = w & 1u;
    const uint32_t i0 = tile_ij[2 * blockIdx.x] * DT, j0 = tile_ij[2 * blockIdx.x + 1] * DT;
    const uint64_t r0 = (uint64_t)blockIdx.y * rows_per_split;
    const uint64_t r1 = min(rows, r0 + rows_per_split);
    d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4){0, 0, 0, 0};
    // Register-staged pipeline: the global loads of block t+1 are issued before the MFMAs of block t and
    // written to LDS after them, so their latency hides under 64 MFMAs per wave.
    // staging map: wave w, trip h -> panel row 4 w + h; lane -> column pair 2 lane: one load instruction reads one
    // row's 1 KB slice as 64 consecutive 16-byte pieces (8 lines; the older map spread 4 rows over 32 half-used lines)
    const uint32_t sc = lane * 2u;
    d2 xa[4], yb[4];
    auto load_block = [&](uint64_t r) {
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const uint64_t gr = r + w * 4u + h;
            const bool rv = gr < r1;
            const uint32_t c = sc;
            xa[h] = (d2){0.0, 0.0};
            yb[h] = (d2){0.0, 0.0};
            if (rv) {
                const double *xr = X + gr * ldx + i0 + c;
                const double *yr = Y + gr * ldy + j0 + c;
                if (i0 + c + 1 < n)
                    xa[h] = *reinterpret_cast<const d2 *>(xr);
                else if (i0 + c < n)
                    xa[h].x = xr[0];
                if (j0 + c + 1 < m)
                    yb[h] = *reinterpret_cast<const d2 *>(yr);
                else if (j0 + c < m)
                    yb[h].x = yr[0];
            }
        }
    };
    auto store_block = [&]() {
#pragma unroll
        for (int h = 0; h < 4; h++) {
            *reinterpret_cast<d2 *>(&As[(w * 4u + h) * DLD + sc]) = xa[h];
            *reinterpret_cast<d2 *>(&Bs[(w * 4u + h) * DLD + sc]) = yb[h];
        }
    };
    if (r0 < r1) {
        load_block(r0);
        store_block();
    }
    __syncthreads();
    for (uint64_t r = r0; r < r1; r += DK) {
        const bool more = r + DK < r1;
        if (more) load_block(r + DK);
        tile_mma(As, Bs, wi, wj, li, lk, acc);
        __syncthreads();
        if (more) store_block();
        __syncthreads();
    }
    double *__restrict__ dst = slab + (size_t)blockIdx.y * n * m;
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int nj = 0; nj < 4; nj++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const uint32_t row = i0 + wi * 64 + mi * 16 + lk + 4 * reg;
                const uint32_t col = j0 + wj * 64 + nj * 16 + li;
                if (row < n && col < m) dst[(size_t)row * m + col] = acc[mi][nj][reg];
            }
}

// ordered sum of the row-split partials; with `symmetric` the strictly-lower tiles were skipped and are mirrored
__global__ void gram_tiled_finish_kernel(const double *__restrict__ slab, uint32_t splits, uint32_t n, uint32_t m,
                                         int symmetric, double *__restrict__ C, const int *__restrict__ skip) {
    if (skip && *skip) return;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)n * m) return;
    uint32_t row = (uint32_t)(e / m), col = (uint32_t)(e % m);
    uint64_t src = e;
    if (symmetric && row / DT > col / DT) src = (uint64_t)col * m + row;
    double s = 0.0;
    for (uint32_t k = 0; k < splits; k++) s += slab[(size_t)k * n * m + src];
    C[e] = s;
}

// Out = beta * Cin + alpha * X W;  grid.x = 128-row tile, grid.y = 128-column tile
__global__ __launch_bounds__(256) void gemm_tiled_kernel(const double *__restrict__ X, uint32_t ldx, uint32_t n,
                                                         const double *__restrict__ W, uint32_t ldw, uint32_t m, uint64_t rows,
                                                         double alpha, double beta, const double *Cin, uint32_t ldc,
                                                         double *Out, uint32_t ldo, const int *__restrict__ skip) {
    if (skip && *skip) return;
    __shared__ double As[DK * DLD];
    __shared__ double Bs[DK * DLD];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t li = lane & 15u, lk = lane >> 4, wi = w >> 1, wj = w & 1u;
    const uint64_t r0 = (uint64_t)blockIdx.x * DT;
    const uint32_t j0 = blockIdx.y * DT;
    d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4){0, 0, 0, 0};
    // A staging: thread -> (rows tid / 8 + 32 h, the k pair tid % 8), written transposed. Eight lanes share a row's 128
    // bytes, so one load instruction touches 8 lines instead of one per lane (the texture path is what the sparse
    // passes running beside these kernels live on); the transposed LDS stores pay an 8-way bank conflict for it,
    // ~6 % of the MFMA time of a step.
    const uint32_t ai = tid >> 3, ak = (tid & 7u) * 2u;
    // B staging: thread -> (k = tid / 16, 8 consecutive columns)
    const uint32_t bk = tid >> 4, bc = (tid & 15u) * 8u;
    d2 av[4], bv[4];
    auto load_block = [&](uint32_t k0) {
        const uint32_t kb = k0 + bk;
        const double *wr = W + (size_t)kb * ldw + j0 + bc;
        const uint32_t kk = k0 + ak;
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const uint64_t arow = r0 + ai + 32u * h;
            av[h] = (d2){0.0, 0.0};
            if (arow < rows) {
                const double *xr = X + arow * ldx + kk;
                if (kk + 1 < n)
                    av[h] = *reinterpret_cast<const d2 *>(xr);
                else if (kk < n)
                    av[h].x = xr[0];
            }
            const uint32_t c = bc + 2u * h;
            bv[h] = (d2){0.0, 0.0};
            if (kb < n) {
                if (j0 + c + 1 < m)
                    bv[h].x = wr[2 * h], bv[h].y = wr[2 * h + 1]; // ldw may be odd: scalar loads
                else if (j0 + c < m)
                    bv[h].x = wr[2 * h];
            }
        }
    };
    auto store_block = [&]() {
#pragma unroll
        for (int h = 0; h < 4; h++) {
            As[ak * DLD + ai + 32u * h] = av[h].x;
            As[(ak + 1u) * DLD + ai + 32u * h] = av[h].y;
            *reinterpret_cast<d2 *>(&Bs[bk * DLD + bc + 2u * h]) = bv[h];
        }
    };
    if (n > 0) {
        load_block(0);
        store_block();
    }
    __syncthreads();
    for (uint32_t k0 = 0; k0 < n; k0 += DK) {
        const bool more = k0 + DK < n;
        if (more) load_block(k0 + DK);
        tile_mma(As, Bs, wi, wj, li, lk, acc);
        __syncthreads();
        if (more) store_block();
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const uint64_t row = r0 + wi * 64 + mi * 16 + lk + 4 * reg;
            if (row >= rows) continue;
#pragma unroll
            for (int nj = 0; nj < 4; nj++) {
                const uint32_t col = j0 + wj * 64 + nj * 16 + li;
                if (col >= m) continue;
                double r = alpha * acc[mi][nj][reg];
                if (beta != 0.0) r = fma(beta, Cin[row * ldc + col], r);
                Out[row * ldo + col] = r;
            }
        }
}

// Narrow results (m <= 64 columns: the Ritz factors U = T W, 10^6 x 500 -> 50): a 128-column tile would spend 61 % of
// its MFMAs on padding. Here the workgroup tile is 256 rows x 64 columns, each of the 4 waves owning 64 rows x 64 columns
// (the same 4 x 4 MFMA tiles per wave); LDS images [k][256 + 16] and [k][64 + 16] keep the bank property of DLD.
constexpr uint32_t SK_R = 256, SK_C = 64, SK_ALD = 272, SK_BLD = 80;
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const double *__restrict__ X, uint32_t ldx, uint32_t n,
                                                          const double *__restrict__ W, uint32_t ldw, uint32_t m, uint64_t rows,
                                                          double alpha, double beta, const double *Cin, uint32_t ldc,
                                                          double *Out, uint32_t ldo, const int *__restrict__ skip) {
    if (skip && *skip) return;
    __shared__ double As[DK * SK_ALD];
    __shared__ double Bs[DK * SK_BLD];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t li = lane & 15u, lk = lane >> 4;
    const uint64_t r0 = (uint64_t)blockIdx.x * SK_R;
    const uint32_t j0 = blockIdx.y * SK_C;
    d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (d4){0, 0, 0, 0};
    // A staging: thread -> (rows tid / 8 + 32 h, the k pair tid % 8): eight lanes share a row's 128 bytes (see gemm_tiled_kernel)
    const uint32_t ai = tid >> 3, ak = (tid & 7u) * 2u;
    // B staging: thread -> (k = tid / 16, 4 consecutive columns)
    const uint32_t bk = tid >> 4, bc = (tid & 15u) * 4u;
    d2 av[8];
    double bv[4];
    auto load_block = [&](uint32_t k0) {
        const uint32_t kk = k0 + ak;
#pragma unroll
        for (int h = 0; h < 8; h++) {
            const uint64_t arow = r0 + ai + 32u * h;
            av[h] = (d2){0.0, 0.0};
            if (arow < rows) {
                const double *xr = X + arow * ldx + kk;
                if (kk + 1 < n)
                    av[h] = *reinterpret_cast<const d2 *>(xr);
                else if (kk < n)
                    av[h].x = xr[0];
            }
        }
        const uint32_t kb = k0 + bk;
        const double *wr = W + (size_t)kb * ldw + j0 + bc;
#pragma unroll
        for (int h = 0; h < 4; h++) bv[h] = (kb < n && j0 + bc + h < m) ? wr[h] : 0.0;
    };
    auto store_block = [&]() {
#pragma unroll
        for (int h = 0; h < 8; h++) {
            As[ak * SK_ALD + ai + 32u * h] = av[h].x;
            As[(ak + 1u) * SK_ALD + ai + 32u * h] = av[h].y;
        }
#pragma unroll
        for (int h = 0; h < 4; h++) Bs[bk * SK_BLD + bc + h] = bv[h];
    };
    if (n > 0) {
        load_block(0);
        store_block();
    }
    __syncthreads();
    for (uint32_t k0 = 0; k0 < n; k0 += DK) {
        const bool more = k0 + DK < n;
        if (more) load_block(k0 + DK);
#pragma unroll
        for (uint32_t kk = 0; kk < DK / 4; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                a[t] = As[(kk * 4 + lk) * SK_ALD + w * 64 + t * 16 + li];
                b[t] = Bs[(kk * 4 + lk) * SK_BLD + t * 16 + li];
            }
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int nj = 0; nj < 4; nj++) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
        }
        __syncthreads();
        if (more) store_block();
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const uint64_t row = r0 + w * 64 + mi * 16 + lk + 4 * reg;
            if (row >= rows) continue;
#pragma unroll
            for (int nj = 0; nj < 4; nj++) {
                const uint32_t col = j0 + nj * 16 + li;
                if (col >= m) continue;
                double r = alpha * acc[mi][nj][reg];
                if (beta != 0.0) r = fma(beta, Cin[row * ldc + col], r);
                Out[row * ldo + col] = r;
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// Small factorizations on the device (one workgroup, the matrix in LDS): the b x b Cholesky factor and its inverse that
// CholeskyQR needs between a Gram kernel and a GEMM. On the host they cost a D2H copy, a stream synchronisation and an H2D
// copy per pass (0.2-0.4 ms each, several per orthonormalisation, and the device idle meanwhile); here the whole
// orthonormalisation is queued without the host looking at anything. Same arithmetic as host_linalg.cpp's chol_upper /
// inv_upper (row-oriented, sums in the same order), same shift rule as orth_cholqr (shifted CholeskyQR, Fukaya et al. 2020).
//   ctl[0]  in/out  1 once this orthonormalisation has converged (or failed): the kernel then returns Rinv = I, so that
//                   the GEMMs queued behind it leave the panel as it is
//   ctl[1]  out     status, sticky: 0 ok, 1 Cholesky failed, 2 non-finite Gram matrix
//   info[0] = max |G - I| of this pass, info[1] = shift used
// check_only: the last queued pass only tests convergence.
constexpr uint32_t CHOL_NMAX = 128;
#ifndef SCANRS_CHOL_TD
#define SCANRS_CHOL_TD 32
#endif
// threads per dimension: 32 x 32 = 1024 threads, 16 waves (4 per SIMD). (16 x 16 = 256 threads, SCANRS_CHOL_TD=16, was built to keep the
// step out of the queue behind the long grids of short workgroups on the projection stream — in a kernel trace it waits 0.5-1.2 ms for a
// CU to empty, profiles/r06e_timeline.txt 134.9 / 166.95 / 199.7 — but runs 314 instead of 200 us at n = 100, and the untraced step is
// the same: 203.2 / 203.5 against 202.8 / 203.0 ms, A/B on one box.)
constexpr uint32_t CHOL_TD = SCANRS_CHOL_TD;
constexpr int CHOL_E = CHOL_NMAX / CHOL_TD;    // entries per thread and dimension: thread (ty, tx) owns (ty + TD a, tx + TD b)
// The matrix lives in REGISTERS (cyclic 32 x 32 distribution, 4 x 4 entries per thread, compile-time indices); a step
// broadcasts one row (and, for the inverse, one column) through LDS and every thread applies the rank-1 update to its own
// entries: two barriers and a few dozen instructions per step. (An LDS-resident version with per-entry read-modify-write
// loops ran 290 us at n = 100, bound by LDS latency; 256 threads with 8 x 8 entries each ran as long — one wave per SIMD
// issues its ~500 instructions per step back to back.)
__global__ __launch_bounds__(CHOL_TD * CHOL_TD) void chol_rinv_kernel(const double *__restrict__ G, uint32_t n, double rows, int pass, int check_only,
                                                        int *__restrict__ ctl, double *__restrict__ Rinv, double *__restrict__ info) {
    __shared__ double rowbuf[CHOL_NMAX], colbuf[CHOL_NMAX], red[3 * CHOL_TD * CHOL_TD];
    const uint32_t tid = threadIdx.x, nt = blockDim.x, ty = tid / CHOL_TD, tx = tid % CHOL_TD;
    // rows of this wave in row block a: wy0 + 32 a and wy0 + 1 + 32 a (a wave is two rows of the thread grid) — uniform tests skip whole blocks
    const uint32_t wy0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) * (64u / CHOL_TD);
    auto identity_out = [&]() {
        if (check_only) return;
        for (uint32_t e = tid; e < n * n; e += nt) Rinv[e] = (e / n == e % n) ? 1.0 : 0.0;
    };
    if (ctl[0]) { // already converged (or failed): nothing to apply
        identity_out();
        return;
    }
    // max |G - I|, largest diagonal entry, finiteness
    double err = 0.0, dmax = 0.0;
    int bad = 0;
    for (uint32_t e = tid; e < n * n; e += nt) {
        const uint32_t i = e / n, j = e - i * n;
        const double g = G[e];
        if (!isfinite(g)) bad = 1;
        err = fmax(err, fabs(g - (i == j ? 1.0 : 0.0)));
        if (i == j) dmax = fmax(dmax, g);
    }
    red[tid] = err;
    red[nt + tid] = dmax;
    red[2 * nt + tid] = (double)bad;
    __syncthreads();
    for (uint32_t s = nt / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] = fmax(red[tid], red[tid + s]);
            red[nt + tid] = fmax(red[nt + tid], red[nt + tid + s]);
            red[2 * nt + tid] = fmax(red[2 * nt + tid], red[2 * nt + tid + s]);
        }
        __syncthreads();
    }
    err = red[0];
    dmax = red[nt];
    bad = red[2 * nt] != 0.0;
    __syncthreads();
    if (tid == 0 && info) info[0] = err;
    if (bad) {
        if (tid == 0) {
            ctl[0] = 1;
            ctl[1] = 2;
        }
        identity_out();
        return;
    }
    if (pass >= 1 && err < 5e-14 * sqrt((double)n)) { // orth_cholqr's stopping rule
        if (tid == 0) ctl[0] = 1;
        identity_out();
        return;
    }
    if (check_only) return; // not converged within the queued passes: ctl[0] stays 0, the host falls back
    double A[CHOL_E][CHOL_E];
    double shift = 0.0;
    for (int tries = 0;; tries++) {
#pragma unroll
        for (int a = 0; a < CHOL_E; a++)
#pragma unroll
            for (int b = 0; b < CHOL_E; b++) {
                const uint32_t i = ty + CHOL_TD * a, j = tx + CHOL_TD * b;
                A[a][b] = (i < n && j < n && j >= i) ? G[i * n + j] + (i == j ? shift : 0.0) : 0.0;
            }
        // G (+ shift I) = R^T R, right-looking: an entry (i, j) loses r_ki r_kj for k = 0, 1, ... — the order of chol_upper's row loop
        bool failed = false;
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t ka = k / CHOL_TD, kt = k % CHOL_TD;
#pragma unroll
            for (int a = 0; a < CHOL_E; a++)
                if ((uint32_t)a == ka && ty == kt) {
#pragma unroll
                    for (int b = 0; b < CHOL_E; b++) rowbuf[tx + CHOL_TD * b] = A[a][b]; // row k as it stands (entries left of the diagonal are zeros)
                }
            __syncthreads();
            const double d = rowbuf[k];
            if (!(d > 0.0) || !isfinite(d)) { // uniform
                failed = true;
                break;
            }
            const double sq = sqrt(d), inv = 1.0 / sq;
            double g[CHOL_E], f[CHOL_E];
#pragma unroll
            for (int b = 0; b < CHOL_E; b++) g[b] = rowbuf[tx + CHOL_TD * b] * inv;
#pragma unroll
            for (int a = 0; a < CHOL_E; a++) f[a] = rowbuf[ty + CHOL_TD * a] * inv;
#pragma unroll
            for (int a = 0; a < CHOL_E; a++) {
                if (wy0 + 64u / CHOL_TD - 1u + CHOL_TD * a < k) continue; // uniform: every row this wave holds in the block is finished
                const uint32_t i = ty + CHOL_TD * a;
#pragma unroll
                for (int b = a; b < CHOL_E; b++) { // blocks left of the diagonal block hold no entry with j >= i
                    const uint32_t j = tx + CHOL_TD * b;
                    if (i == k)
                        A[a][b] = j == k ? sq : (j > k ? g[b] : 0.0);
                    else if (i > k && j >= i)
                        A[a][b] -= f[a] * g[b];
                }
            }
            __syncthreads(); // rowbuf is rewritten by the next step
        }
        __syncthreads();
        if (!failed) break;
        shift = shift == 0.0 ? 11.0 * (rows * n + (double)n * (n + 1)) * 1.1e-16 * dmax : shift * 100.0;
        if (tries + 1 > 12 || !(dmax > 0.0)) {
            if (tid == 0) {
                ctl[0] = 1;
                ctl[1] = 1;
            }
            identity_out();
            return;
        }
    }
    if (tid == 0 && info) info[1] = shift;
    // X = R^-1 in place, rows bottom-up: row m of X is final once rows > m are; it then enters the sums of every row k < m:
    // S[k][j] += r_km x_mj (j >= m). S[k][j] takes the register of r_kj, which was consumed at step j.
    for (int m = (int)n - 1; m >= 0; m--) {
        const uint32_t ma = (uint32_t)m / CHOL_TD, mt = (uint32_t)m % CHOL_TD;
#pragma unroll
        for (int b = 0; b < CHOL_E; b++)
            if ((uint32_t)b == ma && tx == mt) {
#pragma unroll
                for (int a = 0; a < CHOL_E; a++) colbuf[ty + CHOL_TD * a] = A[a][b]; // column m of R (rows < m are used)
            }
        __syncthreads();
        const double invd = 1.0 / colbuf[m];
#pragma unroll
        for (int a = 0; a < CHOL_E; a++)
            if ((uint32_t)a == ma && ty == mt) {
#pragma unroll
                for (int b = 0; b < CHOL_E; b++) {
                    const uint32_t j = tx + CHOL_TD * b;
                    const double x = j == (uint32_t)m ? invd : (j > (uint32_t)m ? -A[a][b] * invd : 0.0);
                    A[a][b] = x;
                    rowbuf[j] = x;
                }
            }
        __syncthreads();
        double cf[CHOL_E], xr[CHOL_E];
#pragma unroll
        for (int a = 0; a < CHOL_E; a++) cf[a] = colbuf[ty + CHOL_TD * a];
#pragma unroll
        for (int b = 0; b < CHOL_E; b++) xr[b] = rowbuf[tx + CHOL_TD * b];
#pragma unroll
        for (int a = 0; a < CHOL_E; a++) {
            if (wy0 + CHOL_TD * a >= (uint32_t)m) continue; // uniform: no row this wave holds in the block is above row m
            const uint32_t k = ty + CHOL_TD * a;
#pragma unroll
            for (int b = a; b < CHOL_E; b++) {
                if (CHOL_TD * b + CHOL_TD - 1u < (uint32_t)m) continue; // uniform: every column of this block is left of column m
                const uint32_t j = tx + CHOL_TD * b;
                if (k < (uint32_t)m && j >= (uint32_t)m) A[a][b] = j == (uint32_t)m ? cf[a] * xr[b] : fma(cf[a], xr[b], A[a][b]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < CHOL_E; a++)
#pragma unroll
        for (int b = 0; b < CHOL_E; b++) {
            const uint32_t i = ty + CHOL_TD * a, j = tx + CHOL_TD * b;
            if (i < n && j < n) Rinv[i * n + j] = j >= i ? A[a][b] : 0.0;
        }
}

bool chol_rinv_ok(uint32_t n) { return n >= 1 && n <= CHOL_NMAX; }
void launch_chol_rinv(Storage &st, const double *G, uint32_t n, uint64_t rows, int pass, bool check_only, int *ctl, double *Rinv, double *info) {
    if (!chol_rinv_ok(n)) fail(SCANRS_ERR_ARGUMENT, "device Cholesky: n out of range");
    if (st.prof.on) st.prof.begin(st.stream, "chol_rinv", (double)n * n * 16.0);
    hipLaunchKernelGGL(chol_rinv_kernel, dim3(1), dim3(CHOL_TD * CHOL_TD), 0, st.stream, G, n, (double)rows, pass, check_only ? 1 : 0, ctl, Rinv, info);
    if (st.prof.on) st.prof.end(st.stream);
    SCANRS_HIP(hipGetLastError());
}

// ctl[1] = 3 (sticky) when max |C| >= limit: the cross-block projection of svd_bk's qr(K) did not reach rounding level
__global__ void absmax_flag_kernel(const double *__restrict__ C, uint32_t count, double limit, int *__restrict__ ctl, double *__restrict__ info) {
    __shared__ double red[256];
    double m = 0.0;
    for (uint32_t e = threadIdx.x; e < count; e += blockDim.x) {
        const double x = fabs(C[e]);
        m = (x > m || x != x) ? x : m;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const double o = red[threadIdx.x + s];
            if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (info) info[0] = red[0];
        ctl[0] = 1; // a check, not an orthonormalisation: complete as soon as it has run
        if (!(red[0] < limit) && ctl[1] == 0) ctl[1] = 3;
    }
}
void launch_absmax_flag(Storage &st, const double *C, uint32_t count, double limit, int *ctl, double *info) {
    hipLaunchKernelGGL(absmax_flag_kernel, dim3(1), dim3(256), 0, st.stream, C, count, limit, ctl, info);
    SCANRS_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
bool gram_tiled_ok(uint32_t n, uint32_t m, uint64_t rows) { return n >= 48 && m >= 48 && rows >= 2048; }
bool gemm_tiled_ok(uint32_t n, uint32_t m, uint64_t rows) { return n >= 16 && m >= 48 && rows >= 2048; }

void launch_gram_tiled(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *Y, uint32_t ldy, uint32_t m,
                       uint64_t rows, double *C) {
    const uint32_t tn = (n + DT - 1) / DT, tm = (m + DT - 1) / DT;
    const bool symmetric = X == Y && ldx == ldy && n == m;
    std::vector<uint32_t> tiles;
    for (uint32_t i = 0; i < tn; i++)
        for (uint32_t j = symmetric ? i : 0; j < tm; j++) {
            tiles.push_back(i);
            tiles.push_back(j);
        }
    const uint32_t n_tiles = (uint32_t)tiles.size() / 2;
    // the list depends on (tn, tm, symmetric) only: uploaded once per shape, so that a Gram product is a pure enqueue
    char tkey[64];
    snprintf(tkey, sizeof tkey, "gramt_tiles_%u_%u_%d", tn, tm, symmetric ? 1 : 0);
    uint32_t *d_tiles = st.scratch.get<uint32_t>(tkey, tiles.size());
    if (st.scratch.filled.insert(tkey).second) {
        SCANRS_HIP(hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice, st.stream));
        SCANRS_SYNC(st.stream);
    }
    // ~2k workgroups, slices of at least 256 rows and a multiple of the staging depth
    uint64_t splits = std::max<uint64_t>(1, std::min<uint64_t>((rows + 255) / 256, (2048 + n_tiles - 1) / n_tiles));
    uint64_t rps = (rows + splits - 1) / splits;
    rps = (rps + DK - 1) / DK * DK;
    splits = std::max<uint64_t>(1, (rows + rps - 1) / rps);
    double *slab = st.scratch.get<double>(st.skey("gram_slab"), (size_t)splits * n * m);
    if (st.prof.on) st.prof.begin(st.stream, "gram_tiled_mfma_f64", (double)rows * (n + (symmetric ? 0 : m)) * 8.0 + (double)n * m * 8.0);
    hipLaunchKernelGGL(gram_tiled_kernel, dim3(n_tiles, (unsigned)splits), dim3(256), 0, st.stream, X, ldx, n, Y, ldy, m, rows, rps,
                       d_tiles, slab, st.skip_flag);
    hipLaunchKernelGGL(gram_tiled_finish_kernel, dim3((unsigned)(((uint64_t)n * m + 255) / 256)), dim3(256), 0, st.stream, slab,
                       (uint32_t)splits, n, m, symmetric ? 1 : 0, C, st.skip_flag);
    if (st.prof.on) st.prof.end(st.stream);
    SCANRS_HIP(hipGetLastError());
}

#include "dense_skinny.inc"

// X W with rows of X aligned for 16-byte loads: straight from memory into the MFMA operands (dense_skinny.inc) — one column group up to
// 112 result columns (svd_bk's panels, the Ritz factors), groups of 64 beyond (svd_rand's 500-column CholeskyQR applications:
// 10^6 x 500 -> 500 in 11.3 ms = 44 TF against 22 ms through the LDS tiles; groups of 112 there: 12.5 ms). Measured against the LDS-tiled
// kernels (profiles/microbench/gemm_skinny_probe.hip): 10^6 x 500 -> 50: 1.50 ms (2.5); 10^6 x 400 -> 100: 2.08 (3.6); 33 k x 400 -> 100:
// 0.11 (0.29); 33 k x 100 -> 100: 0.045 (0.06-0.13).
bool gemm_direct_ok(const double *X, uint32_t ldx, uint32_t n, uint32_t m, uint64_t rows) {
    return m >= 1u && m <= 4096u && n >= 16u && rows >= 64u && ldx % 2u == 0 && (reinterpret_cast<uintptr_t>(X) & 15u) == 0;
}
void launch_gemm_direct(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *W, uint32_t ldw, uint32_t m, uint64_t rows,
                        double alpha, double beta, const double *Cin, uint32_t ldc, double *Out, uint32_t ldo) {
    // up to 112 columns in ONE group (7 MFMA column tiles per wave, X read once: 10^6 x 400 -> 100 in 2.08 ms against 2.40 in two groups of
    // 64); 113-128 columns in two groups
    const uint32_t n_pad = (n + 15u) / 16u * 16u, groups = m <= 112u ? 1u : (m + 63u) / 64u, nt = ((m + 15u) / 16u + groups - 1u) / groups,
                   m_pad = nt * 16u * groups;
    double *Wt = st.scratch.get<double>(st.skey("skinny_wt"), (size_t)n_pad * m_pad);
    if (st.prof.on) st.prof.begin(st.stream, "gemm_skinny_mfma_f64", (double)rows * (n + m) * 8.0 + (double)n * m * 8.0);
    hipLaunchKernelGGL(skinny_wt_kernel, dim3((n_pad * m_pad + 255u) / 256u), dim3(256), 0, st.stream, W, ldw, n, m, n_pad, m_pad, Wt, st.skip_flag);
    const dim3 grid((unsigned)((rows + 64u * SKD_MT - 1) / (64u * SKD_MT)), groups);
#define SCANRS_SKD_LAUNCH(NT)                                                                                                                 \
    hipLaunchKernelGGL(gemm_skinny_direct_kernel<NT>, grid, dim3(256), 0, st.stream, X, ldx, n, Wt, n_pad, m, rows, alpha, beta, Cin, ldc, Out, ldo, \
                       st.skip_flag)
    switch (nt) {
    case 1: SCANRS_SKD_LAUNCH(1); break;
    case 2: SCANRS_SKD_LAUNCH(2); break;
    case 3: SCANRS_SKD_LAUNCH(3); break;
    case 4: SCANRS_SKD_LAUNCH(4); break;
    case 5: SCANRS_SKD_LAUNCH(5); break;
    case 6: SCANRS_SKD_LAUNCH(6); break;
    default: SCANRS_SKD_LAUNCH(7); break;
    }
#undef SCANRS_SKD_LAUNCH
    if (st.prof.on) st.prof.end(st.stream);
    SCANRS_HIP(hipGetLastError());
}

void launch_gemm_tiled(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *W, uint32_t ldw, uint32_t m,
                       uint64_t rows, double alpha, double beta, const double *Cin, uint32_t ldc, double *Out, uint32_t ldo) {
    if ((m - 1u) % DT < SK_C) { // the last (or only) 128-column tile would be at most half full
        if (st.prof.on) st.prof.begin(st.stream, "gemm_skinny_mfma_f64", (double)rows * (n + m) * 8.0 + (double)n * m * 8.0);
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3((unsigned)((rows + SK_R - 1) / SK_R), (m + SK_C - 1) / SK_C), dim3(256), 0, st.stream, X,
                           ldx, n, W, ldw, m, rows, alpha, beta, Cin, ldc, Out, ldo, st.skip_flag);
        if (st.prof.on) st.prof.end(st.stream);
        SCANRS_HIP(hipGetLastError());
        return;
    }
    if (st.prof.on) st.prof.begin(st.stream, "gemm_tiled_mfma_f64", (double)rows * (n + m) * 8.0 + (double)n * m * 8.0);
    if (trace_on()) fprintf(stderr, "[scanrs trace] gemm_tiled rows=%llu n=%u m=%u ldx=%u ldw=%u beta=%g\n", (unsigned long long)rows, n, m, ldx, ldw, beta);
    hipLaunchKernelGGL(gemm_tiled_kernel, dim3((unsigned)((rows + DT - 1) / DT), (m + DT - 1) / DT), dim3(256), 0, st.stream, X, ldx,
                       n, W, ldw, m, rows, alpha, beta, Cin, ldc, Out, ldo, st.skip_flag);
    if (st.prof.on) st.prof.end(st.stream);
    SCANRS_HIP(hipGetLastError());
}

// scanrs_init(): one empty launch per translation unit makes the runtime load this file's code object now instead of inside the
// first real call
__global__ void warm_dense_kernel() {}
void warm_dense(hipStream_t s) { hipLaunchKernelGGL(warm_dense_kernel, dim3(1), dim3(64), 0, s); }

} // namespace scanrs
