// knn.hip — exact k-nearest-neighbours of the PCA scores (scan-rs/src/nn.rs:38-83), SURVEY.md §8 f4.
//
// The reference builds a ball tree (ball_tree crate) over the rows of a cells x d matrix and asks it for the k+1
// nearest points of every row, dropping the row itself. Same result here by exhaustive search: one thread per
// query (its d coordinates in registers), the candidate point wave-uniform in SGPRs (scalar loads), the squared distance as the
// reference forms it (sum of squared differences, nn.rs:14-21 — not the |p|^2 + |q|^2 - 2 p.q expansion, whose
// cancellation would reorder near neighbours), a sorted k-list per thread in private memory. f64 vector FMA bound:
// n_q x n x d multiply-adds. Ties keep ascending index order (the rule of the reference's own test oracle,
// `exhaustive_knn`, nn.rs:112-137; the ball tree's order among exactly equidistant points is a property of that crate).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "common.hpp"

namespace scanrs {

namespace {

constexpr uint32_t KMAX = 128;

template <int DMAX, int THREADS>
__global__ __launch_bounds__(THREADS) void knn_kernel(const double *__restrict__ queries, uint64_t n_q, const double *__restrict__ points,
                                                     uint64_t n_p, uint32_t d, uint32_t k, int skip_same_index,
                                                     uint32_t *__restrict__ out) {
    const uint64_t qi = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    const bool live = qi < n_q;
    double q[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; j++) q[j] = (live && (uint32_t)j < d) ? queries[qi * d + j] : 0.0;
    double best_d[KMAX];
    uint32_t best_i[KMAX];
    uint32_t have = 0;
    double worst = DBL_MAX; // distance a candidate has to beat once the list is full
    // `points` is the (n_p x DMAX) zero-padded copy: the candidate's coordinates are wave-uniform, so they arrive through
    // the scalar cache into SGPRs (s_load_dwordx16) and every lane's VALU work is just subtract + fma per coordinate
    for (uint64_t pi = 0; pi < n_p; pi++) {
        const double *__restrict__ pt = points + pi * DMAX;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < DMAX; j++) {
            const double t = pt[j] - q[j];
            s = fma(t, t, s);
        }
        if (!live || (skip_same_index && pi == qi)) continue;
        if (have == k && !(s < worst)) continue;
        // sorted insertion; equal distances stay in index order because candidates arrive in index order
        uint32_t pos = have < k ? have : k - 1u;
        while (pos > 0 && best_d[pos - 1] > s) {
            best_d[pos] = best_d[pos - 1];
            best_i[pos] = best_i[pos - 1];
            pos--;
        }
        best_d[pos] = s;
        best_i[pos] = (uint32_t)pi;
        if (have < k) have++;
        if (have == k) worst = best_d[k - 1];
    }
    if (!live) return;
    for (uint32_t i = 0; i < k; i++) out[qi * k + i] = i < have ? best_i[i] : 0xFFFFFFFFu; // T::max_value() padding, nn.rs:66
}

__global__ void pad_points_kernel(const double *__restrict__ src, uint32_t ld, uint64_t n, uint32_t d, uint32_t dmax,
                                  double *__restrict__ dst) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * dmax) return;
    const uint64_t r = e / dmax;
    const uint32_t j = (uint32_t)(e % dmax);
    dst[e] = j < d ? src[r * ld + j] : 0.0;
}

template <int DMAX, int THREADS>
void launch(const double *dq, uint32_t ldq, uint64_t n_q, const double *dp, uint32_t ldp, uint64_t n_p, uint32_t d, uint32_t k, int skip,
            uint32_t *dout, hipStream_t s) {
    // zero-padded (n x DMAX) copies: the candidate's coordinates arrive as whole scalar-cache lines, the query's as one run per thread
    DevBuf<double> pp, qp;
    pp.alloc(n_p * DMAX ? n_p * DMAX : 1);
    if (n_p)
        hipLaunchKernelGGL(pad_points_kernel, dim3((unsigned)((n_p * DMAX + 255) / 256)), dim3(256), 0, s, dp, ldp, n_p, d, (uint32_t)DMAX, pp.p);
    const double *q = pp.p;
    if (dq != dp || n_q != n_p || ldq != ldp) {
        qp.alloc(n_q * DMAX);
        hipLaunchKernelGGL(pad_points_kernel, dim3((unsigned)((n_q * DMAX + 255) / 256)), dim3(256), 0, s, dq, ldq, n_q, d, (uint32_t)DMAX, qp.p);
        q = qp.p;
    }
    const dim3 grid((unsigned)((n_q + THREADS - 1) / THREADS)), block(THREADS);
    hipLaunchKernelGGL((knn_kernel<DMAX, THREADS>), grid, block, 0, s, q, n_q, pp.p, n_p, (uint32_t)DMAX, k, skip, dout);
    SCANRS_HIP(hipStreamSynchronize(s)); // the padded copies are released on return
}

} // namespace

void knn_device(const double *d_queries, uint32_t ld_q, uint64_t n_q, const double *d_points, uint32_t ld_p, uint64_t n_p, uint32_t d,
                uint32_t k, bool skip_same_index, uint32_t *out) {
    if (k == 0 || n_q == 0) return;
    if (k > KMAX) fail(SCANRS_ERR_ARGUMENT, "knn: k must not exceed 128");
    if (d == 0 || d > 128) fail(SCANRS_ERR_ARGUMENT, "knn: 1 <= dimensions <= 128");
    if (ld_q < d || ld_p < d) fail(SCANRS_ERR_ARGUMENT, "knn: leading dimension smaller than the number of coordinates");
    if (n_p > 0xFFFFFFFEull) fail(SCANRS_ERR_SHAPE, "knn: too many points for u32 indices");
    DevBuf<uint32_t> dout;
    dout.alloc(n_q * k);
    const int skip = skip_same_index ? 1 : 0;
    if (d <= 8)
        launch<8, 256>(d_queries, ld_q, n_q, d_points, ld_p, n_p, d, k, skip, dout.p, 0);
    else if (d <= 16)
        launch<16, 256>(d_queries, ld_q, n_q, d_points, ld_p, n_p, d, k, skip, dout.p, 0);
    else if (d <= 32)
        launch<32, 256>(d_queries, ld_q, n_q, d_points, ld_p, n_p, d, k, skip, dout.p, 0);
    else if (d <= 52) // top-50 PCA scores, the default of scan-rs-cmd (tools/src/bin/cmd.rs:46-48)
        launch<52, 256>(d_queries, ld_q, n_q, d_points, ld_p, n_p, d, k, skip, dout.p, 0);
    else if (d <= 64)
        launch<64, 256>(d_queries, ld_q, n_q, d_points, ld_p, n_p, d, k, skip, dout.p, 0);
    else
        launch<128, 64>(d_queries, ld_q, n_q, d_points, ld_p, n_p, d, k, skip, dout.p, 0);
    SCANRS_HIP(hipGetLastError());
    SCANRS_HIP(hipMemcpy(out, dout.p, n_q * k * 4, hipMemcpyDeviceToHost));
}

// queries (n_q x d) against points (n_p x d), both row-major host arrays; out n_q x k.
void knn_host(const double *queries, uint64_t n_q, const double *points, uint64_t n_p, uint32_t d, uint32_t k, bool skip_same_index,
              uint32_t *out) {
    if (k == 0 || n_q == 0) return;
    DevBuf<double> dq, dp;
    const bool same = queries == points && n_q == n_p;
    dp.alloc(n_p * d ? n_p * d : 1);
    if (n_p) SCANRS_HIP(hipMemcpy(dp.p, points, n_p * d * 8, hipMemcpyHostToDevice));
    if (!same) {
        dq.alloc(n_q * d ? n_q * d : 1);
        SCANRS_HIP(hipMemcpy(dq.p, queries, n_q * d * 8, hipMemcpyHostToDevice));
    }
    knn_device(same ? dp.p : dq.p, d, n_q, dp.p, d, n_p, d, k, skip_same_index, out);
}

} // namespace scanrs
