// Small dense factorizations on the host (the k x k side of the solvers). They stand where the
// reference calls LAPACK through ndarray-linalg on matrices of order <= 5b (scan-rs/src/dim_red/
// bk_svd.rs:94-139): Cholesky + triangular inverse for the CholeskyQR panels, and a symmetric
// eigensolver (Householder tridiagonalisation + implicit-shift QL) for the projected Gram matrix.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>
#include <unistd.h>
#include <vector>

#include "host_team.hpp"
#include "common.hpp"

namespace scanrs {

bool chol_upper(double *g, int n) {
    // row-oriented: R[i][j], j >= i, G = R^T R
    for (int i = 0; i < n; i++) {
        double *ri = g + (size_t)i * n;
        for (int k = 0; k < i; k++) {
            const double *rk = g + (size_t)k * n;
            const double f = rk[i];
            if (f != 0.0)
                for (int j = i; j < n; j++) ri[j] -= f * rk[j];
        }
        const double d = ri[i];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        const double s = std::sqrt(d), inv = 1.0 / s;
        ri[i] = s;
        for (int j = i + 1; j < n; j++) ri[j] *= inv;
        for (int j = 0; j < i; j++) ri[j] = 0.0;
    }
    return true;
}

void inv_upper(double *r, int n) {
    // X = R^{-1}, upper triangular; solve R X = I column block by row recurrences, bottom-up.
    std::vector<double> x((size_t)n * n, 0.0);
    for (int i = n - 1; i >= 0; i--) {
        double *xi = x.data() + (size_t)i * n;
        const double *ri = r + (size_t)i * n;
        const double inv = 1.0 / ri[i];
        xi[i] = 1.0;
        for (int k = i + 1; k < n; k++) {
            const double f = ri[k];
            if (f == 0.0) continue;
            const double *xk = x.data() + (size_t)k * n;
            for (int j = k; j < n; j++) xi[j] -= f * xk[j];
        }
        for (int j = i; j < n; j++) xi[j] *= inv;
    }
    std::copy(x.begin(), x.end(), r);
}

// Symmetric eigenproblem A = Z diag(w) Z^T.
// 1) Householder reduction A -> T (tridiagonal), reflectors applied row-wise so every inner loop is
//    unit-stride; 2) accumulate Q; 3) implicit QL with Wilkinson shifts on (d, e), rotating rows of Q^T.
bool sym_eig(const double *a_in, int n, double *w, double *z) {
    if (n == 0) return true;
    std::vector<double> A(a_in, a_in + (size_t)n * n);
    std::vector<double> d(n), e(n, 0.0), tau(n, 0.0);
    std::vector<double> V((size_t)n * n, 0.0); // reflector k stored in row k (entries k+1..n-1)
    std::vector<double> p(n), v(n);

    for (int k = 0; k < n - 2; k++) {
        // x = A[k+1.., k] (use row k by symmetry)
        double *ak = A.data() + (size_t)k * n;
        double scale = 0.0;
        for (int i = k + 1; i < n; i++) scale = std::max(scale, std::fabs(ak[i]));
        if (scale == 0.0) {
            tau[k] = 0.0;
            e[k] = 0.0;
            continue;
        }
        double nrm2 = 0.0;
        for (int i = k + 1; i < n; i++) {
            v[i] = ak[i] / scale;
            nrm2 += v[i] * v[i];
        }
        const double alpha = v[k + 1];
        double beta = std::sqrt(nrm2);
        if (alpha > 0) beta = -beta;
        // H = I - tau v v^T with v[k+1] = 1
        const double v0 = alpha - beta;
        tau[k] = (beta - alpha) / beta;
        for (int i = k + 2; i < n; i++) v[i] /= v0;
        v[k + 1] = 1.0;
        e[k] = beta * scale;
        // p = tau * A22 v
        for (int i = k + 1; i < n; i++) {
            const double *ai = A.data() + (size_t)i * n;
            double s = 0.0;
            for (int j = k + 1; j < n; j++) s += ai[j] * v[j];
            p[i] = tau[k] * s;
        }
        double pv = 0.0;
        for (int i = k + 1; i < n; i++) pv += p[i] * v[i];
        const double half = 0.5 * tau[k] * pv;
        for (int i = k + 1; i < n; i++) p[i] -= half * v[i]; // w
        for (int i = k + 1; i < n; i++) {
            double *ai = A.data() + (size_t)i * n;
            const double vi = v[i], pi = p[i];
            for (int j = k + 1; j < n; j++) ai[j] -= vi * p[j] + pi * v[j];
        }
        double *vk = V.data() + (size_t)k * n;
        for (int i = k + 1; i < n; i++) vk[i] = v[i];
    }
    for (int i = 0; i < n; i++) d[i] = A[(size_t)i * n + i];
    if (n >= 2) e[n - 2] = A[(size_t)(n - 2) * n + (n - 1)];

    // Qt = (H_0 H_1 ... H_{n-3})^T, built by applying reflectors to the identity from the last to the first.
    // We keep Qt row-major where row j of Qt is eigenvector-basis column j of Q, i.e. Qt[j][i] = Q[i][j].
    std::vector<double> Qt((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) Qt[(size_t)i * n + i] = 1.0;
    // Q = H_0 ... H_{n-3}; Q^T = H_{n-3} ... H_0. Apply to columns: Q^T <- Q^T (built as product), do
    // Q <- H_k Q for k = n-3..0 on Q stored as rows of Q (row-major Q), then transpose into Qt.
    {
        std::vector<double> Q((size_t)n * n, 0.0);
        for (int i = 0; i < n; i++) Q[(size_t)i * n + i] = 1.0;
        std::vector<double> s(n);
        for (int k = n - 3; k >= 0; k--) {
            if (tau[k] == 0.0) continue;
            const double *vk = V.data() + (size_t)k * n;
            // s = v^T Q (over rows k+1..n-1), Q -= tau v s^T
            std::fill(s.begin(), s.end(), 0.0);
            for (int i = k + 1; i < n; i++) {
                const double vi = vk[i];
                const double *qi = Q.data() + (size_t)i * n;
                for (int j = k + 1; j < n; j++) s[j] += vi * qi[j];
            }
            for (int i = k + 1; i < n; i++) {
                const double f = tau[k] * vk[i];
                double *qi = Q.data() + (size_t)i * n;
                for (int j = k + 1; j < n; j++) qi[j] -= f * s[j];
            }
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Qt[(size_t)j * n + i] = Q[(size_t)i * n + j];
    }

    // implicit QL; e[i] couples d[i] and d[i+1]
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; l++) {
        int iter = 0;
        int m;
        do {
            for (m = l; m < n - 1; m++) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= eps * dd) break;
            }
            if (m != l) {
                if (iter++ == 200) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, pp = 0.0;
                int i;
                for (i = m - 1; i >= l; i--) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    r = std::hypot(f, g);
                    e[i + 1] = r;
                    if (r == 0.0) {
                        d[i + 1] -= pp;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - pp;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    pp = s * r;
                    d[i + 1] = g + pp;
                    g = c * r - b;
                    double *zi = Qt.data() + (size_t)i * n;
                    double *zi1 = Qt.data() + (size_t)(i + 1) * n;
                    for (int k2 = 0; k2 < n; k2++) {
                        f = zi1[k2];
                        zi1[k2] = s * zi[k2] + c * f;
                        zi[k2] = c * zi[k2] - s * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= pp;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    // sort descending
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int x, int y) { return d[x] > d[y]; });
    for (int j = 0; j < n; j++) {
        w[j] = d[order[j]];
        const double *src = Qt.data() + (size_t)order[j] * n;
        for (int i = 0; i < n; i++) z[(size_t)i * n + j] = src[i];
    }
    return true;
}


// The k largest eigenvalues of tridiag(e, d, e), descending, by bisection on Sturm counts (LAPACK's dstebz idea): the count
// recurrence q_i = (d_i - x) - e_{i-1}^2 / q_{i-1} is one dependent division per row and shift — the k intervals are advanced
// together, so the inner loop runs over k independent shifts and vectorises. 60 halvings x n rows x k shifts: 0.4 ms at
// n = 500, k = 50 against 2.6 ms for the implicit QL sweep over ALL eigenvalues; accuracy eps * |T| like QL.
static void tridiag_topk_bisect(const double *d, const double *e, int n, int k, double *out) {
    double glo = d[0], ghi = d[0], e2max = 0.0;
    for (int i = 0; i < n; i++) { // Gershgorin
        const double r = (i ? std::fabs(e[i - 1]) : 0.0) + (i < n - 1 ? std::fabs(e[i]) : 0.0);
        glo = std::min(glo, d[i] - r);
        ghi = std::max(ghi, d[i] + r);
        if (i < n - 1) e2max = std::max(e2max, e[i] * e[i]);
    }
    const double span = std::max(std::fabs(glo), std::fabs(ghi));
    glo -= 2.0 * 2.220446049250313e-16 * span + 1e-300;
    ghi += 2.0 * 2.220446049250313e-16 * span + 1e-300;
    const double pivmin = 2.2250738585072014e-308 * std::max(1.0, e2max);
    std::vector<double> lo(k, glo), hi(k, ghi), mid(k), q(k), cnt(k), e2(n, 0.0);
    for (int i = 1; i < n; i++) e2[i] = e[i - 1] * e[i - 1];
    for (int it = 0; it < 200; it++) {
        bool all_done = true;
        for (int t = 0; t < k; t++) {
            mid[t] = 0.5 * (lo[t] + hi[t]);
            cnt[t] = 0.0;
            q[t] = 1.0;
            all_done = all_done && !(mid[t] > lo[t] && mid[t] < hi[t]); // the interval has no interior point left
        }
        if (all_done) break;
        double *__restrict__ qq = q.data();
        double *__restrict__ cc = cnt.data();
        const double *__restrict__ mm = mid.data();
        for (int i = 0; i < n; i++) {
            const double di = d[i], ei = e2[i];
            for (int t = 0; t < k; t++) {
                double x = (di - mm[t]) - ei / qq[t]; // e2[0] = 0: the first row is d_0 - x
                x = std::fabs(x) < pivmin ? -pivmin : x;
                qq[t] = x;
                cc[t] += x < 0.0 ? 1.0 : 0.0; // eigenvalues below the shift
            }
        }
        for (int t = 0; t < k; t++) {
            // the t-th largest eigenvalue is the (n - t)-th smallest: it lies below mid iff at least n - t eigenvalues do
            if (cnt[t] >= (double)(n - t))
                hi[t] = mid[t];
            else
                lo[t] = mid[t];
        }
    }
    for (int t = 0; t < k; t++) out[t] = 0.5 * (lo[t] + hi[t]);
}

// Top-k eigenpairs of a symmetric matrix: Householder tridiagonalisation, all eigenvalues by implicit QL
// on (d, e) alone (O(n^2)), the k leading eigenvectors by inverse iteration on the tridiagonal matrix with
// re-orthogonalisation against the vectors already found, then back-transformation through the stored
// reflectors. Same decomposition as sym_eig restricted to k columns at ~1/4 of the flops (the Rayleigh-Ritz
// step of the solvers only returns the first k triplets, scan-rs/src/dim_red/bk_svd.rs:106-108,135-137).
bool sym_eig_topk(const double *a_in, int n, int k, double *w, double *z) {
    if (n == 0 || k == 0) return true;
    if (k > n) return false;
    static const bool tr = getenv("SCANRS_TRACE_EIG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!tr) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[eig] %-20s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<double> A(a_in, a_in + (size_t)n * n);
    // Work on A / 2^e with 2^e ~ max |a_ij| (exact: a power of two) and hand the eigenvalues back times 2^e: the squared norms of
    // the inverse iteration underflow for matrices of entries below ~1e-150 (and overflow above ~1e150) otherwise.
    double amax = 0.0;
    for (double x : A) amax = std::max(amax, std::fabs(x));
    if (!std::isfinite(amax)) return false;
    if (amax == 0.0) { // the zero matrix: every orthonormal set is an eigenbasis
        for (int j = 0; j < k; j++) w[j] = 0.0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < k; j++) z[(size_t)i * k + j] = i == j ? 1.0 : 0.0;
        return true;
    }
    int scale_exp = 0;
    if (amax > 0.0) {
        (void)std::frexp(amax, &scale_exp);
        if (scale_exp != 0)
            for (double &x : A) x = std::ldexp(x, -scale_exp);
    }
    struct Rescale { // on every way out
        double *w;
        int k, e;
        ~Rescale() {
            if (e != 0)
                for (int j = 0; j < k; j++) w[j] = std::ldexp(w[j], e);
        }
    } rescale{w, k, scale_exp};
    // The reduction touches the lower triangle only: packed row by row (row i: columns 0 .. i, rows starting on 64-byte lines) it is
    // half the footprint of the square array — 1 MB instead of 2 MB at n = 500, the size of the one core's L2 that every step streams
    // it through (round 6: 5.0 -> see DESIGN 4c).
    std::vector<size_t> roff(n + 1, 0);
    for (int i = 0; i < n; i++) roff[i + 1] = roff[i] + (((size_t)i + 1 + 7) & ~(size_t)7);
    std::vector<double> Ap(roff[n] + 8, 0.0);
    double *const ap0 = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(Ap.data()) + 63) & ~(uintptr_t)63);
    auto row = [&](int i) -> double * { return ap0 + roff[i]; };
    for (int i = 0; i < n; i++) {
        const double *src = A.data() + (size_t)i * n;
        double *dst = row(i);
        for (int j = 0; j <= i; j++) dst[j] = src[j];
    }
    std::vector<double>().swap(A);
    std::vector<double> d(n), e(n, 0.0), tau(n, 0.0);
    std::vector<double> V((size_t)n * n, 0.0);
    std::vector<double> p(n), v(n), vn(n), pn(n);
    // Householder tridiagonalisation on the LOWER triangle only, one pass over the trailing block per step: the
    // rank-2 update of step kk (A -= v p^T + p v^T) and the product p' = A' v' of step kk+1 share the pass — column
    // kk+1 is updated first (it defines v'), then every row is updated and at once folded into p' (dot for its own
    // entry, axpy for the entries of the rows above it). 8 n^2 bytes of traffic per step instead of 24 n^2.
    auto make_reflector = [&](int kk, const double *colv /* entries kk+1..n-1 of column kk, indexed by row */, double *vv) -> bool {
        double scale = 0.0;
        for (int i = kk + 1; i < n; i++) scale = std::max(scale, std::fabs(colv[i]));
        if (scale == 0.0) {
            tau[kk] = 0.0;
            e[kk] = 0.0;
            return false;
        }
        double nrm2 = 0.0;
        for (int i = kk + 1; i < n; i++) {
            vv[i] = colv[i] / scale;
            nrm2 += vv[i] * vv[i];
        }
        const double alpha = vv[kk + 1];
        double beta = std::sqrt(nrm2);
        if (alpha > 0) beta = -beta;
        const double v0 = alpha - beta;
        tau[kk] = (beta - alpha) / beta;
        for (int i = kk + 2; i < n; i++) vv[i] /= v0;
        vv[kk + 1] = 1.0;
        e[kk] = beta * scale;
        return true;
    };
    // finish p: p = tau (A v) - (tau/2)(p.v) v, entries kk+1..n-1
    auto finish_p = [&](int kk, double *pp, const double *vv) {
        for (int j = kk + 1; j < n; j++) pp[j] *= tau[kk];
        double pv = 0.0;
        for (int i = kk + 1; i < n; i++) pv += pp[i] * vv[i];
        const double half = 0.5 * tau[kk] * pv;
        for (int i = kk + 1; i < n; i++) pp[i] -= half * vv[i];
    };
    std::vector<double> colbuf(n);
    bool have = false; // a reflector (v, p) of step kk is pending application
    if (n > 2) {
        for (int i = 1; i < n; i++) colbuf[i] = row(i)[0];
        have = make_reflector(0, colbuf.data(), v.data());
        if (have) { // p = A22 v from the lower triangle
            for (int j = 1; j < n; j++) p[j] = 0.0;
            for (int i = 1; i < n; i++) {
                const double *__restrict__ ai = row(i);
                double *__restrict__ pp = p.data();
                const double vi = v[i];
                double s = 0.0;
                {
#pragma clang fp reassociate(on)
                    for (int j = 1; j < i; j++) {
                        s += ai[j] * v[j];
                        pp[j] += ai[j] * vi;
                    }
                }
                pp[i] += s + ai[i] * vi;
            }
            finish_p(0, p.data(), v.data());
            double *vk = V.data();
            for (int i = 1; i < n; i++) vk[i] = v[i];
        }
    }
    // Steps kk = 0 .. n-3. The pass over the trailing rows is cut into VS = 4 fixed slots — row i belongs to slot
    // (i - c - 1) % 4 — each folding its rows into ITS OWN partial of p' (the scatter half of the symmetric product would
    // race otherwise); the partials are added in slot order after the pass. The arithmetic is therefore the same whether
    // one thread walks the four slots or a team of 2 / 4 shares them: results do not depend on the thread count, the
    // host, or on timing (replicated ranks must get identical factors). With a team (n >= 768) there are four spin barriers
    // per step; thread 0 keeps the serial parts (column update, reflector, finishing p'). If the barriers turn out slow
    // (an oversubscribed or CPU-throttled host: the others are not running), the team is dismissed and thread 0 goes on alone.
    constexpr int VS = 4;
    const int T_env = global_options().eig_threads >= 4 ? 4 : (global_options().eig_threads >= 2 ? 2 : 1);
    // measured on the MI355X host (256 cores): n = 1000: 48.6 -> 31 ms with 4 threads; n = 500: 6.1 -> 7.8 ms (the 2 MB matrix
    // lives in one core's L2 and the three barriers per step cost more than the split saves; round 6, another box: 6.9 -> 9.8 (2 threads) /
    // 11.5 (4)) -> team from n = 768 on
    int T = (n >= 768 && std::thread::hardware_concurrency() >= 8) ? T_env : 1;
    HostTeam *pool = T > 1 ? HostTeam::acquire(T) : nullptr;
    if (!pool) T = 1;
    std::vector<std::vector<double>> pn_part(VS - 1, std::vector<double>(n, 0.0)); // slot 0 uses pn itself
    struct SpinBarrier {
        std::atomic<int> count{0}, gen{0};
        int n_threads = 1;
        void wait() {
            if (n_threads == 1) return;
            const int g = gen.load(std::memory_order_acquire);
            if (count.fetch_add(1, std::memory_order_acq_rel) == n_threads - 1) {
                count.store(0, std::memory_order_relaxed);
                gen.store(g + 1, std::memory_order_release);
            } else {
                int spins = 0;
                while (gen.load(std::memory_order_acquire) == g) {
                    if (++spins < 20000)
                        __builtin_ia32_pause();
                    else
                        std::this_thread::yield();
                }
            }
        }
    } bar;
    bar.n_threads = T;
    // shared step state, written by thread 0 between the barriers
    int step_c = 0;
    bool step_have = false, step_have_next = false, team_done = false;
    auto slot_pass = [&](int slot) {
        const int c = step_c;
        const bool hv = step_have, hn = step_have_next;
        if (!hv && !hn) return;
        double *__restrict__ pq = slot == 0 ? pn.data() : pn_part[slot - 1].data();
        if (hn)
            for (int j = c + 1; j < n; j++) pq[j] = 0.0;
        int i = c + 1 + ((slot - (c + 1)) % VS + VS) % VS; // rows with i % VS == slot: a row never changes owner
        // Two rows of the slot at a time, update and product fused: the loop is bound by its loads and stores (row, p, v, v', p'
        // per element: 6 loads + 2 stores per row and element in two separate loops), and a pair shares everything but its own
        // rows (6 loads + 3 stores per TWO elements). Same operations on the same operands in the same order as row by row:
        // p'[j] takes row i's term before row i + VS's.
        if (hv && hn) {
            const double *__restrict__ pp = p.data();
            const double *__restrict__ vv = v.data();
            const double *__restrict__ vq = vn.data();
            for (; i + VS < n; i += 2 * VS) {
                const int i2 = i + VS;
                double *__restrict__ a1 = row(i);
                double *__restrict__ a2 = row(i2);
                const double v1 = v[i], p1 = p[i], v2 = v[i2], p2 = p[i2], w1 = vq[i], w2 = vq[i2];
                double s1 = 0.0, s2 = 0.0;
                for (int j = c + 1; j < i; j++) {
                    const double x1 = a1[j] - (v1 * pp[j] + p1 * vv[j]);
                    const double x2 = a2[j] - (v2 * pp[j] + p2 * vv[j]);
                    a1[j] = x1;
                    a2[j] = x2;
                    {
#pragma clang fp reassociate(on)
                        s1 += x1 * vq[j];
                        s2 += x2 * vq[j];
                    }
                    double t = pq[j];
                    t += x1 * w1;
                    t += x2 * w2;
                    pq[j] = t;
                }
                a1[i] -= v1 * pp[i] + p1 * vv[i];
                pq[i] += s1 + a1[i] * w1;
                for (int j = i; j < i2; j++) { // what row i + VS has beyond row i's diagonal
                    const double x2 = a2[j] - (v2 * pp[j] + p2 * vv[j]);
                    a2[j] = x2;
                    {
#pragma clang fp reassociate(on)
                        s2 += x2 * vq[j];
                    }
                    pq[j] += x2 * w2;
                }
                a2[i2] -= v2 * pp[i2] + p2 * vv[i2];
                pq[i2] += s2 + a2[i2] * w2;
            }
        }
        for (; i < n; i += VS) {
            double *__restrict__ ai = row(i);
            if (hv) {
                const double vi = v[i], pi = p[i];
                const double *__restrict__ pp = p.data();
                const double *__restrict__ vv = v.data();
                for (int j = c + 1; j <= i; j++) ai[j] -= vi * pp[j] + pi * vv[j];
            }
            if (hn) {
                const double *__restrict__ vq = vn.data();
                const double vi = vq[i];
                double s_ = 0.0;
                {
#pragma clang fp reassociate(on)
                    for (int j = c + 1; j < i; j++) {
                        s_ += ai[j] * vq[j];
                        pq[j] += ai[j] * vi;
                    }
                }
                pq[i] += s_ + ai[i] * vi;
            }
        }
    };
    // the pending reflector applied to column c, by the owners of the rows (thread 0 never touches their cache lines):
    // the updated entries also go to colbuf, from which thread 0 builds the next reflector
    auto col_pass = [&](int slot) {
        const int c = step_c;
        for (int i = c + ((slot - c) % VS + VS) % VS; i < n; i += VS) {
            double &a = row(i)[c];
            if (step_have) a -= v[i] * p[c] + p[i] * v[c];
            colbuf[i] = a;
        }
    };
    const int T_team = T;
    if (pool)
        pool->start(T_team, [&](int t) {
            for (;;) {
                bar.wait(); // step state (c, have) published
                if (team_done) return;
                for (int slot = t; slot < VS; slot += T_team) col_pass(slot);
                bar.wait(); // column c complete
                bar.wait(); // reflector published
                for (int slot = t; slot < VS; slot += T_team) slot_pass(slot);
                bar.wait(); // partials complete
            }
        });
    double waited_ms = 0.0;
    for (int kk = 0; kk < n - 2; kk++) {
        // apply the pending reflector of step kk to column kk+1 and the diagonal entry (kk+1, kk+1)
        const int c = kk + 1;
        step_c = c;
        step_have = have;
        bool have_next = false;
        auto since = [](std::chrono::steady_clock::time_point t0_) {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count();
        };
        if (T > 1) {
            auto w0 = std::chrono::steady_clock::now();
            bar.wait();
            double w = kk > 0 ? since(w0) : 0.0; // the first barrier includes waking the team
            for (int slot = 0; slot < VS; slot += T_team) col_pass(slot);
            w0 = std::chrono::steady_clock::now();
            bar.wait();
            w += since(w0);
            if (c < n - 2) have_next = make_reflector(c, colbuf.data(), vn.data());
            step_have_next = have_next;
            bar.wait();
            // one pass over rows c+1..n-1, columns c+1..i: update with (v, p), accumulate pn = A' vn
            for (int slot = 0; slot < VS; slot += T_team) slot_pass(slot);
            w0 = std::chrono::steady_clock::now();
            bar.wait();
            waited_ms += w + since(w0);
            if (kk >= 7 && waited_ms > 0.05 * (kk + 1) + 0.3) { // > 50 us per step waiting for the others: they are not running
                team_done = true;
                bar.wait();
                pool->join();
                pool->release();
                pool = nullptr;
                T = 1;
                if (tr) fprintf(stderr, "[eig] team dismissed after %d steps (%.2f ms spent waiting)\n", kk + 1, waited_ms);
            }
        } else {
            for (int slot = 0; slot < VS; slot++) col_pass(slot);
            if (c < n - 2) have_next = make_reflector(c, colbuf.data(), vn.data());
            step_have_next = have_next;
            for (int slot = 0; slot < VS; slot++) slot_pass(slot);
        }
        if (have_next) {
            for (int slot = 1; slot < VS; slot++) {
                const double *__restrict__ part = pn_part[slot - 1].data();
                double *__restrict__ pq = pn.data();
                for (int j = c + 1; j < n; j++) pq[j] += part[j];
            }
            finish_p(c, pn.data(), vn.data());
            double *vk = V.data() + (size_t)c * n;
            for (int i = c + 1; i < n; i++) vk[i] = vn[i];
            std::swap(v, vn);
            std::swap(p, pn);
        }
        have = have_next;
    }
    if (pool) {
        team_done = true;
        bar.wait();
        pool->join();
        pool->release();
        pool = nullptr;
        if (tr) fprintf(stderr, "[eig] team of %d: %.3f ms spent waiting at barriers\n", T, waited_ms);
    }
    for (int i = 0; i < n; i++) d[i] = row(i)[i];
    if (n >= 2) e[n - 2] = row(n - 1)[n - 2];
    const std::vector<double> td = d, te = e; // keep T; QL below destroys its copy
    lap("tridiagonalise");

    // eigenvalues only
    const double eps = 2.220446049250313e-16;
    const bool bisect = (size_t)k * 4 <= (size_t)n;
    if (bisect) tridiag_topk_bisect(td.data(), te.data(), n, k, d.data());
    for (int l = 0; l < n && !bisect; l++) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; m++) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= eps * dd) break;
            }
            if (m != l) {
                if (iter++ == 200) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::sqrt(g * g + 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, pp = 0.0;
                int i;
                for (i = m - 1; i >= l; i--) {
                    const double f = s * e[i], b = c * e[i];
                    r = std::sqrt(f * f + g * g);
                    e[i + 1] = r;
                    if (r == 0.0) {
                        d[i + 1] -= pp;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - pp;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    pp = s * r;
                    d[i + 1] = g + pp;
                    g = c * r - b;
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= pp;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    if (!bisect) std::sort(d.begin(), d.end(), [](double x, double y) { return x > y; });
    lap(bisect ? "bisection (top k)" : "QL eigenvalues");
    double tnorm = 0.0;
    for (int i = 0; i < n; i++) tnorm = std::max(tnorm, std::fabs(td[i]) + (i ? std::fabs(te[i - 1]) : 0.0) + (i < n - 1 ? std::fabs(te[i]) : 0.0));
    if (tnorm == 0.0) tnorm = 1.0;

    // inverse iteration on T = tridiag(te, td, te)
    std::vector<double> X((size_t)k * n, 0.0); // row j = eigenvector j of T
    std::vector<double> dl(n), dd(n), du(n), du2(n), x(n);
    std::vector<int> piv(n);
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    double prev_shift = 0.0;
    for (int j = 0; j < k; j++) {
        w[j] = d[j];
        double shift = d[j];
        // keep shifts of (numerically) repeated eigenvalues apart so each solve is well posed
        if (j > 0 && std::fabs(shift - prev_shift) < 10.0 * eps * tnorm) shift = prev_shift - 10.0 * eps * tnorm;
        prev_shift = shift;
        // LU of T - shift*I with partial pivoting (tridiagonal: two super-diagonals after pivoting)
        for (int i = 0; i < n; i++) {
            dd[i] = td[i] - shift;
            du[i] = i < n - 1 ? te[i] : 0.0;
            dl[i] = i < n - 1 ? te[i] : 0.0;
            du2[i] = 0.0;
        }
        for (int i = 0; i < n - 1; i++) {
            if (std::fabs(dd[i]) >= std::fabs(dl[i])) {
                piv[i] = 0;
                if (dd[i] == 0.0) dd[i] = eps * tnorm;
                const double f = dl[i] / dd[i];
                dl[i] = f;
                dd[i + 1] -= f * du[i];
            } else {
                piv[i] = 1;
                const double f = dd[i] / dl[i];
                dd[i] = dl[i];
                dl[i] = f;
                const double t = du[i];
                du[i] = dd[i + 1];
                dd[i + 1] = t - f * dd[i + 1];
                if (i < n - 2) {
                    du2[i] = du[i + 1];
                    du[i + 1] = -f * du[i + 1];
                }
            }
        }
        if (dd[n - 1] == 0.0) dd[n - 1] = eps * tnorm;
        for (int i = 0; i < n; i++) { // deterministic pseudo-random start
            rng ^= rng << 13;
            rng ^= rng >> 7;
            rng ^= rng << 17;
            x[i] = ((double)(rng >> 11) / 9007199254740992.0) - 0.5;
        }
        double *xj = X.data() + (size_t)j * n;
        int q_lo = j;
        while (q_lo > 0 && std::fabs(d[q_lo - 1] - d[j]) <= 3e-2 * tnorm) q_lo--;
        for (int it = 0; it < 5; it++) {
            // solve L U y = P x
            for (int i = 0; i < n - 1; i++) {
                if (piv[i]) std::swap(x[i], x[i + 1]);
                x[i + 1] -= dl[i] * x[i];
            }
            x[n - 1] /= dd[n - 1];
            if (n >= 2) x[n - 2] = (x[n - 2] - du[n - 2] * x[n - 1]) / dd[n - 2];
            for (int i = n - 3; i >= 0; i--) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / dd[i];
            // orthogonalise against the accepted vectors whose eigenvalue lies within 3e-2 |T| of this one (twice), normalise. Inverse
            // iteration leaves a component eps |T| / |w_i - w_j| of a converged neighbour i in vector j: beyond that distance it is
            // below 1e-14 by itself (4e-13 at the residual bound of the test below) (LAPACK's dstein draws the line at 1e-3 |T|); the eigenvalues are sorted, so the neighbours are
            // the vectors q_lo .. j-1. (Round 6: against ALL accepted vectors this loop was 1.0 of the 1.3 ms of the 50 vectors of a
            // 500-row problem.)
            for (int rep = 0; rep < 2; rep++)
                for (int q2 = q_lo; q2 < j; q2++) {
                    const double *xq = X.data() + (size_t)q2 * n;
                    double dot = 0.0;
                    {
#pragma clang fp reassociate(on)
                        for (int i = 0; i < n; i++) dot += xq[i] * x[i];
                    }
                    for (int i = 0; i < n; i++) x[i] -= dot * xq[i];
                }
            double nrm = 0.0;
            {
#pragma clang fp reassociate(on)
                for (int i = 0; i < n; i++) nrm += x[i] * x[i];
            }
            nrm = std::sqrt(nrm);
            if (!(nrm > 0.0) || !std::isfinite(nrm)) return false;
            for (int i = 0; i < n; i++) x[i] /= nrm;
            if (it >= 2) { // residual test ||T x - w x|| <= tol
                double res = 0.0;
                for (int i = 0; i < n; i++) {
                    double r = (td[i] - d[j]) * x[i];
                    if (i) r += te[i - 1] * x[i - 1];
                    if (i < n - 1) r += te[i] * x[i + 1];
                    res = std::max(res, std::fabs(r));
                }
                if (res <= 50.0 * eps * tnorm) break;
            }
        }
        for (int i = 0; i < n; i++) xj[i] = x[i];
    }
    lap("inverse iteration");
    // back-transform: Z = H_0 H_1 ... H_{n-3} X. Every vector goes through the reflectors on its own (a dot product and an update per
    // reflector, both over contiguous arrays), so the k vectors are dealt over a team of up to four host threads when there is enough of
    // them to pay for waking it (1.27 -> 0.45 ms at n = 500, k = 50 on the MI355X host); the arithmetic of a vector does not depend on
    // who runs it, nor on how many run: results are identical with and without the team.
    {
        auto back = [&](int j0, int j1) {
            // four vectors at a time share the pass over a reflector
            for (int jb = j0; jb < j1; jb += 4) {
                const int nv = std::min(4, j1 - jb);
                double *xs[4];
                for (int t = 0; t < 4; t++) xs[t] = X.data() + (size_t)(jb + std::min(t, nv - 1)) * n;
                for (int kk = n - 3; kk >= 0; kk--) {
                    const double tk_ = tau[kk];
                    if (tk_ == 0.0) continue;
                    const double *__restrict__ vk = V.data() + (size_t)kk * n;
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    {
#pragma clang fp reassociate(on)
                        for (int i = kk + 1; i < n; i++) {
                            const double vi = vk[i];
                            s0 += vi * xs[0][i];
                            s1 += vi * xs[1][i];
                            s2 += vi * xs[2][i];
                            s3 += vi * xs[3][i];
                        }
                    }
                    s0 *= tk_, s1 *= tk_, s2 *= tk_, s3 *= tk_;
                    if (nv == 4) {
                        for (int i = kk + 1; i < n; i++) {
                            const double vi = vk[i];
                            xs[0][i] -= vi * s0;
                            xs[1][i] -= vi * s1;
                            xs[2][i] -= vi * s2;
                            xs[3][i] -= vi * s3;
                        }
                    } else {
                        const double ss[4] = {s0, s1, s2, s3};
                        for (int t = 0; t < nv; t++)
                            for (int i = kk + 1; i < n; i++) xs[t][i] -= vk[i] * ss[t];
                    }
                }
            }
        };
        const int groups = (k + 3) / 4; // units of four vectors
        int Tb = 1;
        if ((size_t)n * k >= 16384 && groups >= 2 && std::thread::hardware_concurrency() >= 8)
            Tb = std::min(global_options().eig_threads >= 4 ? 4 : (global_options().eig_threads >= 2 ? 2 : 1), groups);
        HostTeam *team = Tb > 1 ? HostTeam::acquire(Tb) : nullptr;
        if (!team) Tb = 1;
        auto share = [&](int t) { // thread t: groups [g0, g1)
            const int g0 = (int)((long)groups * t / Tb), g1 = (int)((long)groups * (t + 1) / Tb);
            back(g0 * 4, std::min(k, g1 * 4));
        };
        if (team) {
            team->start(Tb, [&](int t) { share(t); });
            share(0);
            team->join();
            team->release();
        } else {
            back(0, k);
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < k; j++) z[(size_t)i * k + j] = X[(size_t)j * n + i];
    }
    lap("back-transform");
    return true;
}

} // namespace scanrs
