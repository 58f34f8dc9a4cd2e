// Truncated-SVD drivers on device panels: svd_bk, svd_rand, irlba of scan-rs/src/dim_red/.
//
// The reference is generic over `T: DataMat + Dot<...>` (dim_red/bk_svd.rs:41-46); here T is the device
// handle and the two products are mat_apply(.., transpose = false / true, ..). Panels that the reference
// keeps as (b x m) row-major are held transposed, long-dimension-major (m x b), because that is the layout
// the gather product reads one coalesced row per nonzero from.
//
// Dense steps: `.qr()?.0` (dgeqrf + dorgqr through ndarray-linalg) becomes block Gram-Schmidt +
// CholeskyQR with re-orthogonalisation — any orthonormal basis of the same column space gives the same
// Rayleigh-Ritz values, see DESIGN.md — and `svddc_into` of the 5b x n projection becomes an
// eigendecomposition of its 5b x 5b Gram matrix (only the top k triplets are returned by the reference).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <sys/mman.h>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"
#include "host_team.hpp"

namespace scanrs {

// ---- rand-family generator for the seeded start panel ("parity unpinned") -----
struct SmallRng {
    uint64_t s[4];
    explicit SmallRng(uint64_t seed) {
        uint64_t state = seed;
        for (int i = 0; i < 4; i++) {
            state += 0x9E3779B97F4A7C15ull;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[i] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    double uniform_m1_1() { // Uniform::new(-1.0, 1.0): [1,2) mantissa trick, then * 2 + (-1)
        const uint64_t bits = (next() >> 12) | 0x3FF0000000000000ull;
        double v12;
        memcpy(&v12, &bits, 8);
        return (v12 - 1.0) * 2.0 + (-1.0);
    }
    double normal() { // Box-Muller on the same stream (the reference's ziggurat is not reproduced)
        double u1, u2;
        do {
            u1 = ((double)(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        } while (u1 <= 0.0);
        u2 = ((double)(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

// ---- GF(2) jump tables for the device-side panel generator (kernels.hip, omega_jump_kernel) ---------------------
namespace {
struct BitMat { // 256 x 256 over GF(2); row i = mask of the input bits that feed output bit i
    uint64_t r[256][4];
};
void bm_mul(const BitMat &a, const BitMat &b, BitMat &c) { // c = a after b
    for (int i = 0; i < 256; i++) {
        uint64_t acc[4] = {0, 0, 0, 0};
        for (int w = 0; w < 4; w++) {
            uint64_t bits = a.r[i][w];
            while (bits) {
                const int k = __builtin_ctzll(bits) + 64 * w;
                bits &= bits - 1;
                acc[0] ^= b.r[k][0];
                acc[1] ^= b.r[k][1];
                acc[2] ^= b.r[k][2];
                acc[3] ^= b.r[k][3];
            }
        }
        for (int w = 0; w < 4; w++) c.r[i][w] = acc[w];
    }
}
void bm_identity(BitMat &m) {
    memset(&m, 0, sizeof(m));
    for (int i = 0; i < 256; i++) m.r[i][i >> 6] = 1ull << (i & 63);
}
void xoshiro_step_matrix(BitMat &t) { // one state transition of xoshiro256++ (the output scrambler is not part of it)
    memset(&t, 0, sizeof(t));
    for (int j = 0; j < 256; j++) {
        uint64_t s[4] = {0, 0, 0, 0};
        s[j >> 6] = 1ull << (j & 63);
        const uint64_t tt = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= tt;
        s[3] = (s[3] << 45) | (s[3] >> 19);
        for (int i = 0; i < 256; i++)
            if ((s[i >> 6] >> (i & 63)) & 1ull) t.r[i][j >> 6] |= 1ull << (j & 63);
    }
}
} // namespace

// J^(2^k), k = 0..n_pow-1 with J = T^d, flattened [k][256][4]
static std::vector<uint64_t> jump_powers(uint64_t d, int n_pow) {
    auto t = std::make_unique<BitMat>(), acc = std::make_unique<BitMat>(), tmp = std::make_unique<BitMat>(),
         sq = std::make_unique<BitMat>();
    xoshiro_step_matrix(*t);
    bm_identity(*acc);
    *sq = *t;
    for (uint64_t e = d; e; e >>= 1) { // acc = T^d
        if (e & 1) {
            bm_mul(*sq, *acc, *tmp);
            *acc = *tmp;
        }
        if (e >> 1) {
            bm_mul(*sq, *sq, *tmp);
            *sq = *tmp;
        }
    }
    std::vector<uint64_t> out((size_t)n_pow * 256 * 4);
    for (int k = 0; k < n_pow; k++) {
        memcpy(out.data() + (size_t)k * 1024, acc->r, sizeof(acc->r));
        if (k + 1 < n_pow) {
            bm_mul(*acc, *acc, *tmp);
            *acc = *tmp;
        }
    }
    return out;
}

// The tables depend on (d, n_pow) only and cost ~10 ms of one host core: computed once per process, by a background thread that
// the first handle creation starts (scanrs_mat_create*), for n_pow up to JUMP_NPOW_SHARED — panels of up to 2^26 x 512 draws; a
// prefix of the table serves every smaller panel.
constexpr uint64_t JUMP_D = 512;
constexpr int JUMP_NPOW_SHARED = 26;
namespace {
std::once_flag g_jump_once;
std::shared_future<std::vector<uint64_t>> g_jump_future;
}
void jump_tables_prefetch() {
    std::call_once(g_jump_once, [] { g_jump_future = std::async(std::launch::async, [] { return jump_powers(JUMP_D, JUMP_NPOW_SHARED); }).share(); });
}

// Fill a device panel with the seeded Uniform(-1, 1) stream: `out` holds the (seq_rows x seq_cols) row-major
// sequence, or its transpose, with leading dimension ld.
static void omega_fill_device(Storage &st, uint64_t seed, uint64_t seq_rows, uint64_t seq_cols, double *out, uint32_t ld,
                              bool transpose) {
    const uint64_t total = seq_rows * seq_cols;
    if (total == 0) return;
    const uint64_t d = 512; // draws per device stream
    const uint64_t streams = (total + d - 1) / d;
    int n_pow = 1;
    while ((1ull << n_pow) < streams) n_pow++;
    if (st.jump_d != d || st.jump_npow < n_pow) { // tables depend only on (d, n_pow): uploaded once per handle
        std::vector<uint64_t> own;
        const std::vector<uint64_t> *tab = &own;
        if (d == JUMP_D && n_pow <= JUMP_NPOW_SHARED) {
            jump_tables_prefetch();
            tab = &g_jump_future.get(); // [k][256][4]: the first n_pow planes are the table of a smaller panel
        } else {
            own = jump_powers(d, n_pow);
        }
        const size_t count = (size_t)n_pow * 1024;
        st.jump_tab.alloc(count);
        SCANRS_HIP(hipMemcpyAsync(st.jump_tab.p, tab->data(), count * 8, hipMemcpyHostToDevice, st.stream));
        SCANRS_SYNC(st.stream);
        st.jump_d = d;
        st.jump_npow = n_pow;
    }
    SmallRng rng(seed);
    launch_omega_jump(st, st.jump_tab.p, n_pow, rng.s, d, total, out, ld, seq_cols, transpose);
}

void omega_fill(uint64_t seed, uint64_t count, double *out) {
    SmallRng rng(seed);
    for (uint64_t i = 0; i < count; i++) out[i] = rng.uniform_m1_1();
}

// ---- snoop -------------------------------------------------------------------------------------------------
static void progress_check(const scanrs_snoop *sn, double p) {
    // CancelProgress::set_progress_check (snoop/src/lib.rs:45-57)
    if (!sn) return;
    if (sn->cancel && __atomic_load_n(sn->cancel, __ATOMIC_RELAXED)) fail(SCANRS_ERR_CANCELLED, "cancellation error");
    if (sn->progress) sn->progress(sn->ctx, p);
}

// ---- panel helpers --------------------------------------------------------------------------------------------
struct Ctx {
    scanrs_mat *m;
    Storage &st;
    hipStream_t s;
    CurrentHandle cur; // a timed-out wait inside the solver names this handle's streams
    // a solver repeats the same products many times: lets the auto path build the hybrid product's tile layout up front
    explicit Ctx(scanrs_mat *mm) : m(mm), st(*mm->st), s(mm->st->stream), cur(mm->st.get()) { st.tile_hint++; }
    ~Ctx() { st.tile_hint--; }
    Ctx(const Ctx &) = delete;
    Ctx &operator=(const Ctx &) = delete;
    double *dev(const char *key, size_t count) { return st.scratch.get<double>(key, count); }
    void sync() { SCANRS_SYNC(s); }
    void h2d(double *d, const double *h, size_t n) {
        SCANRS_HIP(hipMemcpyAsync(d, h, n * 8, hipMemcpyHostToDevice, s));
        sync(); // host staging buffers are pageable and reused
    }
    void d2h(double *h, const double *d, size_t n) { SCANRS_D2H(h, d, n * 8, s); }
};

// Everything launched through the handle inside the scope goes to `other` (kernels, scratch zero-fills, events);
// the host thread is the only user of the handle, so this is a plain save / restore.
struct StreamSwap {
    Storage &st;
    hipStream_t saved, saved_scratch;
    StreamSwap(Storage &s, hipStream_t other) : st(s), saved(s.stream), saved_scratch(s.scratch.stream) {
        st.stream = other;
        st.scratch.stream = other;
    }
    ~StreamSwap() {
        st.stream = saved;
        st.scratch.stream = saved_scratch;
    }
};

// upload a compact host matrix (rows x l) into a padded device panel
static void upload_panel(Ctx &c, const double *h, uint64_t rows, uint32_t l, double *d, uint32_t ld) {
    if (rows == 0 || l == 0) return;
    SCANRS_HIP(hipMemcpy2DAsync(d, (size_t)ld * 8, h, (size_t)l * 8, (size_t)l * 8, rows, hipMemcpyHostToDevice, c.s));
    c.sync();
}
// Large factors (the cell-side PcaResult panel is 400 MB at 10^6 x 50) go to the caller's pageable array through a
// ring of pinned slots: the DMA engine fills slot i + 1 at link speed while host threads copy slot i out — a
// direct copy into pageable memory is staged by the runtime on one thread (5.7 GB/s measured, 70 ms of a 580 ms call).
// `gate(first_row, end_row)` (optional) is called before the copy of those rows is queued: the caller makes `cs` wait for whatever
// still produces them (ritz_finish: the row block of the last GEMM, so that the copy of finished blocks runs beside the later ones).
static void download_panel_staged(Ctx &c, const double *d, uint32_t ld, uint64_t rows, uint32_t l, double *h, hipStream_t cs = nullptr,
                                  const std::function<void(uint64_t, uint64_t)> &gate = nullptr) {
    if (!cs) cs = c.s;
    constexpr int NS = 8;            // slots in flight
    constexpr size_t SLOT = 8u << 20; // bytes per slot
    const size_t row_bytes = (size_t)l * 8;
    const uint64_t slot_rows = std::max<uint64_t>(1, SLOT / row_bytes);
    const size_t n_chunks = (size_t)((rows + slot_rows - 1) / slot_rows);
    char *stage = (char *)c.st.pinned((size_t)NS * slot_rows * row_bytes);
    // The caller's array is usually fresh (run_pca hands out new arrays: untouched pages): with 4 KB pages the copy below takes
    // 10^5 page faults per 400 MB; where transparent huge pages are to be asked for (THP "madvise"), ask — 200 faults instead.
    {
        const uintptr_t a = ((uintptr_t)h + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), e = ((uintptr_t)h + (size_t)rows * row_bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
    }
    const unsigned T = std::max(1u, std::min(c.st.d2h_threads, std::thread::hardware_concurrency()));
    hipEvent_t ev[NS];
    for (auto &e : ev) SCANRS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::atomic<size_t> ready{0};
    std::atomic<bool> stop{false};
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[n_chunks]);
    for (size_t i = 0; i < n_chunks; i++) done[i].store(0, std::memory_order_relaxed);
    auto chunk_rows = [&](size_t i) { return std::min<uint64_t>(slot_rows, rows - (uint64_t)i * slot_rows); };
    auto worker = [&](unsigned t) {
        for (size_t i = t; i < n_chunks; i += T) {
            while (ready.load(std::memory_order_acquire) <= i) {
                if (stop.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            memcpy((char *)h + (size_t)i * slot_rows * row_bytes, stage + (i % NS) * slot_rows * row_bytes, chunk_rows(i) * row_bytes);
            done[i].store(1, std::memory_order_release);
        }
    };
    // the parked host team if nobody else has it (creating a thread costs 0.1-3 ms, four of them per download and two downloads per
    // PCA: round 6), fresh threads otherwise (a second shard's download at the same time)
    HostTeam *team = T <= 8u ? HostTeam::acquire((int)T + 1) : nullptr; // (the team's threads are never destroyed: not for a caller that asks for hundreds)
    std::vector<std::thread> workers;
    if (team)
        team->start((int)T + 1, [&](int t) { worker((unsigned)t - 1u); });
    else
        for (unsigned t = 0; t < T; t++) workers.emplace_back([&, t] { worker(t); });
    auto join_all = [&] {
        if (team) {
            team->join();
            team->release();
            team = nullptr;
        }
        for (auto &w : workers) w.join();
        workers.clear();
        for (auto &e : ev) (void)hipEventDestroy(e);
    };
    try {
        for (size_t i = 0; i < n_chunks; i++) {
            if (i >= NS)
                while (!done[i - NS].load(std::memory_order_acquire)) std::this_thread::yield();
            if (gate) gate((uint64_t)i * slot_rows, (uint64_t)i * slot_rows + chunk_rows(i));
            if (ld == l) // contiguous rows: a plain copy (the DMA engine's fast path; the pitched form runs at about half its rate)
                SCANRS_HIP(hipMemcpyAsync(stage + (i % NS) * slot_rows * row_bytes, d + (size_t)i * slot_rows * ld, chunk_rows(i) * row_bytes,
                                          hipMemcpyDeviceToHost, cs));
            else
                SCANRS_HIP(hipMemcpy2DAsync(stage + (i % NS) * slot_rows * row_bytes, row_bytes, d + (size_t)i * slot_rows * ld, (size_t)ld * 8,
                                            row_bytes, chunk_rows(i), hipMemcpyDeviceToHost, cs));
            SCANRS_HIP(hipEventRecord(ev[i % NS], cs));
            if (i >= 1) {
                SCANRS_SYNC_EVENT(ev[(i - 1) % NS]);
                ready.store(i, std::memory_order_release);
            }
        }
        SCANRS_SYNC(cs);
        ready.store(n_chunks, std::memory_order_release);
    } catch (...) {
        stop.store(true);
        join_all();
        throw;
    }
    join_all();
}
static void download_panel(Ctx &c, const double *d, uint32_t ld, uint64_t rows, uint32_t l, double *h) {
    if (rows == 0 || l == 0) return;
    if (rows * (uint64_t)l * 8 >= (8u << 20)) return download_panel_staged(c, d, ld, rows, l, h);
    SCANRS_D2H_2D(h, d, (size_t)ld * 8, (size_t)l * 8, rows, c.s);
}

// Gram matrix of two panels that live on the same side; reduced across ranks when that side is sharded.
static void gram_host(Ctx &c, const double *X, uint32_t ldx, uint32_t n, const double *Y, uint32_t ldy, uint32_t m,
                      uint64_t rows, bool sharded_rows, std::vector<double> &out) {
    double *dC = c.dev("gram_out", (size_t)n * m);
    launch_gram(c.st, X, ldx, n, Y, ldy, m, rows, dC);
    if (sharded_rows) allreduce_f64(c.st, dC, (uint64_t)n * m);
    out.resize((size_t)n * m);
    c.d2h(out.data(), dC, out.size());
}

// Out = beta * Out + alpha * X * W with a host W (n x m, row-major)
static void gemm_hostw(Ctx &c, const double *X, uint32_t ldx, uint32_t n, const std::vector<double> &W, uint32_t m,
                       uint64_t rows, double alpha, double beta, double *Out, uint32_t ldo, const char *wkey = "gemm_w") {
    double *dW = c.dev(wkey, (size_t)n * m);
    c.h2d(dW, W.data(), (size_t)n * m);
    launch_gemm_nn(c.st, X, ldx, n, dW, m, m, rows, alpha, beta, Out, ldo, Out, ldo);
}

// Last resort of the orthonormalisations, for panels whose columns are numerically DEPENDENT (a matrix of rank below the
// panel width: CholeskyQR, shifted or not, cannot converge on them): modified Gram-Schmidt with re-orthogonalisation on the
// host; a column that is dependent on its predecessors is replaced by a random direction orthogonal to them — what the
// Householder QR of the reference (`.qr()`, bk_svd.rs:94,98,123,127) leaves in such a column is as arbitrary, and any
// orthonormal completion gives the same Rayleigh-Ritz values. The panel then no longer is (original basis) * C, so the
// coefficient bookkeeping of svd_bk is void: Storage::orth_fallbacks tells the solver to compute the projection directly.
static void orth_host_mgs(Ctx &c, double *P, uint32_t ld, uint32_t n, uint64_t rows) {
    Tick tk("  orth_host_mgs (rank-deficient panel)");
    std::vector<double> X((size_t)rows * n);
    download_panel(c, P, ld, rows, n, X.data());
    // column-major copy: the inner loops run over contiguous columns
    std::vector<double> Q((size_t)n * rows);
    for (uint64_t r = 0; r < rows; r++)
        for (uint32_t j = 0; j < n; j++) Q[(size_t)j * rows + r] = X[(size_t)r * n + j];
    SmallRng rng(0x5ca9a5d1u + n);
    auto dot = [&](const double *a, const double *b) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        uint64_t r = 0;
        for (; r + 4 <= rows; r += 4) {
            s0 += a[r] * b[r];
            s1 += a[r + 1] * b[r + 1];
            s2 += a[r + 2] * b[r + 2];
            s3 += a[r + 3] * b[r + 3];
        }
        for (; r < rows; r++) s0 += a[r] * b[r];
        return (s0 + s1) + (s2 + s3);
    };
    for (uint32_t j = 0; j < n; j++) {
        double *v = Q.data() + (size_t)j * rows;
        const double norm0 = std::sqrt(dot(v, v));
        for (int attempt = 0; attempt < 8; attempt++) {
            double before = std::sqrt(dot(v, v));
            for (int pass = 0; pass < 2; pass++)
                for (uint32_t i = 0; i < j; i++) {
                    const double *qi = Q.data() + (size_t)i * rows;
                    const double h = dot(qi, v);
                    for (uint64_t r = 0; r < rows; r++) v[r] -= h * qi[r];
                }
            const double after = std::sqrt(dot(v, v));
            // dependent (or zero) column: what is left is rounding noise of the projections
            const bool dependent = attempt == 0 ? !(after > 1e-10 * norm0) || !(norm0 > 0.0) : !(after > 1e-3 * before);
            if (!dependent && std::isfinite(after)) {
                const double inv = 1.0 / after;
                for (uint64_t r = 0; r < rows; r++) v[r] *= inv;
                break;
            }
            if (attempt == 7) fail(SCANRS_ERR_NUMERICAL, "orthonormalisation: no independent direction found");
            for (uint64_t r = 0; r < rows; r++) v[r] = rng.normal();
        }
    }
    for (uint64_t r = 0; r < rows; r++)
        for (uint32_t j = 0; j < n; j++) X[(size_t)r * n + j] = Q[(size_t)j * rows + r];
    upload_panel(c, X.data(), rows, n, P, ld);
    c.st.orth_fallbacks++;
}

static void orth_cholqr(Ctx &c, double *P, double *tmp, uint32_t ld, uint32_t n, uint64_t rows, bool sharded_rows,
                        std::vector<double> *coef = nullptr, uint32_t coef_rows = 0, bool may_complete = true);

// The same completion for a panel whose rows are SHARDED over the ranks (svd_rand's range finder on the cell side) or too large
// to take to the host: everything is decided from the all-reduced Gram matrix, which every rank holds bit for bit, so all ranks
// take the same branch and meet in the same collectives. G = Z diag(w) Z^T; the directions with w_i > 1e-12 w_1 are kept as
// Q1 = P Z_r diag(w_r)^-1/2 (orthonormal up to eps sigma_1 / sigma_i), the others are replaced by columns of a fixed
// pseudo-random function of (global row, column) — every rank fills its own rows — made orthogonal to Q1 by two projections and
// to each other by CholeskyQR; one more CholeskyQR over the whole panel polishes. Like the host Gram-Schmidt this voids
// Q = K C (Storage::orth_fallbacks).
static void orth_gram_complete(Ctx &c, double *P, double *tmp, uint32_t ld, uint32_t n, uint64_t rows, bool sharded_rows) {
    Tick tk("  orth_gram_complete (rank-deficient panel, decided from the reduced Gram matrix)");
    std::vector<double> G;
    gram_host(c, P, ld, n, P, ld, n, rows, sharded_rows, G);
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t j = i + 1; j < n; j++) {
            if (!std::isfinite(G[(size_t)i * n + j])) fail(SCANRS_ERR_NUMERICAL, "orthonormalisation: non-finite Gram matrix");
            const double a = 0.5 * (G[(size_t)i * n + j] + G[(size_t)j * n + i]);
            G[(size_t)i * n + j] = G[(size_t)j * n + i] = a;
        }
    std::vector<double> w(n), Z((size_t)n * n);
    if (!sym_eig(G.data(), (int)n, w.data(), Z.data())) fail(SCANRS_ERR_NUMERICAL, "orthonormalisation: eigensolver did not converge");
    uint32_t r = 0;
    while (r < n && w[r] > 1e-12 * w[0] && w[r] > 0.0) r++;
    // Q1 = P W1, W1 = Z[:, :r] diag(w)^-1/2 (n x r), into tmp[:, 0:r]
    if (r) {
        std::vector<double> W1((size_t)n * r);
        for (uint32_t i = 0; i < n; i++)
            for (uint32_t j = 0; j < r; j++) W1[(size_t)i * r + j] = Z[(size_t)i * n + j] / std::sqrt(w[j]);
        double *dW = c.dev("orth_gc_w", (size_t)n * r);
        c.h2d(dW, W1.data(), W1.size());
        launch_gemm_nn(c.st, P, ld, n, dW, r, r, rows, 1.0, 0.0, nullptr, 0, tmp, ld);
    }
    const uint32_t nr = n - r;
    if (nr) {
        const uint64_t row0 = sharded_rows ? c.st.shard.outer_begin : 0;
        const uint32_t ldr = even_up(nr);
        double *R = c.dev("orth_gc_r", (size_t)rows * ldr), *Rtmp = c.dev("orth_gc_rtmp", (size_t)rows * ldr); // the random block, compact
        launch_fill_hash(c.st, R, ldr, rows, row0, 0, nr, 0x5ca9a5d1ull + n);
        for (int pass = 0; pass < 2 && r; pass++) { // R -= Q1 (Q1^T R)
            std::vector<double> C;
            gram_host(c, tmp, ld, r, R, ldr, nr, rows, sharded_rows, C);
            gemm_hostw(c, tmp, ld, r, C, nr, rows, -1.0, 1.0, R, ldr, "orth_gc_c");
        }
        orth_cholqr(c, R, Rtmp, ldr, nr, rows, sharded_rows, nullptr, 0, false); // within itself (well conditioned: CholeskyQR converges)
        launch_copy_cols(c.st, R, ldr, tmp + r, ld, rows, nr);
    }
    SCANRS_HIP(hipMemcpyAsync(P, tmp, (size_t)rows * ld * 8, hipMemcpyDeviceToDevice, c.s));
    orth_cholqr(c, P, tmp, ld, n, rows, sharded_rows, nullptr, 0, false); // polish: the kept directions are orthonormal to eps sigma_1 / sigma_r only
    c.st.orth_fallbacks++;
}

// Orthonormalise the columns of P (rows x n, ld) in place: iterated CholeskyQR, with a diagonal shift
// when the Gram matrix is numerically singular (shifted CholeskyQR). tmp: same size as P.
// coef (optional, coef_rows x n row-major): kept equal to the matrix C with P = (original basis) * C, i.e.
// every right-multiplication applied to P is applied to it too.
static void orth_cholqr(Ctx &c, double *P, double *tmp, uint32_t ld, uint32_t n, uint64_t rows, bool sharded_rows,
                        std::vector<double> *coef, uint32_t coef_rows, bool may_complete) {
    Tick tk("  orth_cholqr");
    // what takes over when CholeskyQR cannot converge (a rank-deficient panel): Gram-Schmidt on the host for a replicated panel of
    // moderate size, the Gram-matrix route for a sharded or a very large one
    auto complete = [&]() {
        if (!may_complete) fail(SCANRS_ERR_NUMERICAL, "orthonormalisation did not converge");
        if (!sharded_rows && rows * (uint64_t)n <= (1ull << 26)) return orth_host_mgs(c, P, ld, n, rows);
        return orth_gram_complete(c, P, tmp, ld, n, rows, sharded_rows);
    };
    std::vector<double> G, R;
    for (int pass = 0; pass < 8; pass++) {
        gram_host(c, P, ld, n, P, ld, n, rows, sharded_rows, G);
        double err = 0.0, dmax = 0.0;
        for (uint32_t i = 0; i < n; i++)
            for (uint32_t j = 0; j < n; j++) {
                const double g = G[(size_t)i * n + j];
                if (!std::isfinite(g)) fail(SCANRS_ERR_NUMERICAL, "orthonormalisation: non-finite Gram matrix");
                err = std::max(err, std::fabs(g - (i == j ? 1.0 : 0.0)));
                if (i == j) dmax = std::max(dmax, g);
            }
        if (pass >= 1 && err < 5e-14 * std::sqrt((double)n)) return;
        R = G;
        double shift = 0.0;
        int tries = 0;
        while (!chol_upper(R.data(), (int)n)) {
            // shifted CholeskyQR (Fukaya et al. 2020): G + s I, s ~ 11 (rows n + n(n+1)) u ||X||^2
            shift = shift == 0.0 ? 11.0 * ((double)rows * n + (double)n * (n + 1)) * 1.1e-16 * dmax : shift * 100.0;
            if (++tries > 12 || !(dmax > 0.0)) return complete();
            R = G;
            for (uint32_t i = 0; i < n; i++) R[(size_t)i * n + i] += shift;
        }
        inv_upper(R.data(), (int)n);
        if (coef) { // C <- C * R^-1 (R^-1 upper triangular)
            Tick tk2("    coef Rinv update");
            std::vector<double> nc((size_t)coef_rows * n, 0.0);
            for (uint32_t r = 0; r < coef_rows; r++) {
                const double *__restrict__ cr = coef->data() + (size_t)r * n;
                double *__restrict__ o = nc.data() + (size_t)r * n;
                bool any = false;
                for (uint32_t p2 = 0; p2 < n; p2++) any = any || cr[p2] != 0.0;
                if (!any) continue;
                for (uint32_t p2 = 0; p2 < n; p2++) {
                    const double x = cr[p2];
                    const double *__restrict__ rr = R.data() + (size_t)p2 * n;
                    for (uint32_t j = p2; j < n; j++) o[j] += x * rr[j];
                }
            }
            coef->swap(nc);
        }
        double *dW = c.dev("orth_w", (size_t)n * n);
        c.h2d(dW, R.data(), (size_t)n * n);
        launch_gemm_nn(c.st, P, ld, n, dW, n, n, rows, 1.0, 0.0, nullptr, 0, tmp, ld);
        SCANRS_HIP(hipMemcpyAsync(P, tmp, (size_t)rows * ld * 8, hipMemcpyDeviceToDevice, c.s));
    }
    return complete();
}

// Orthonormalise block `Bj` (rows x b, ld ldb) against the first `nprev` columns of Q (ld ldq) and
// within itself; rounds of (project, CholeskyQR) until the projection is at rounding level.
// coef / cfull (optional): coefficient bookkeeping, Bj = Korig * coef (coef: q x b), Q = Korig * cfull (q x q).
static void orth_against(Ctx &c, const double *Q, uint32_t ldq, uint32_t nprev, double *Bj, double *tmp, uint32_t ldb,
                         uint32_t b, uint64_t rows, bool sharded_rows, std::vector<double> *coef = nullptr,
                         const std::vector<double> *cfull = nullptr, uint32_t q = 0) {
    std::vector<double> C;
    for (int round = 0; round < 4; round++) {
        if (nprev) {
            gram_host(c, Q, ldq, nprev, Bj, ldb, b, rows, sharded_rows, C);
            double cmax = 0.0;
            for (double x : C) cmax = std::max(cmax, std::fabs(x));
            if (round >= 1 && cmax < 1e-14) return;
            gemm_hostw(c, Q, ldq, nprev, C, b, rows, -1.0, 1.0, Bj, ldb);
            if (coef) { // coef -= cfull[:, :nprev] * C   (cfull is block upper triangular)
                Tick tk2("    coef projection update");
                for (uint32_t r = 0; r < nprev; r++) {
                    double *__restrict__ o = coef->data() + (size_t)r * b;
                    const double *__restrict__ cf = cfull->data() + (size_t)r * q;
                    for (uint32_t p2 = (r / b) * b; p2 < nprev; p2++) {
                        const double x = cf[p2];
                        const double *__restrict__ cc = C.data() + (size_t)p2 * b;
                        for (uint32_t j = 0; j < b; j++) o[j] -= x * cc[j];
                    }
                }
            }
        }
        orth_cholqr(c, Bj, tmp, ldb, b, rows, sharded_rows, coef, q);
        if (!nprev) return;
    }
}

// ---- the same orthonormalisations without a host round trip (Storage::device_factor) -------------------------------------
// Every CholeskyQR pass is Gram kernel -> chol_rinv_kernel (one workgroup: stopping rule, Cholesky with the shift rule,
// inverse) -> GEMM, all queued; a converged orthonormalisation turns its remaining passes into products with the identity.
// The verdicts (converged? Cholesky failed? non-finite?) are left in a control block per call and read ONCE, with the
// coefficient matrix, when the Krylov basis is complete: anything but "converged, ok" makes svd_bk start over on the host
// path (rank-deficient inputs that need more shifted passes than are queued here).
struct DevOrth {
    int *ctl = nullptr;     // 2 ints per call: [done, status]
    double *info = nullptr; // 2 doubles per call: [max |G - I| or max |C| of the last check, shift]
    uint32_t used = 0, cap = 0;
    // svd_bk draws one slot per iteration (the panel's CholeskyQR) and three per Krylov block (two rounds + the final check):
    // 4 n_iter - 3 in all — the blocks are sized from n_iter (the reference puts no bound on it, bk_svd.rs:16-53)
    void init(Ctx &c, uint32_t slots) {
        cap = slots;
        ctl = c.st.scratch.get<int>("orth_ctl", 2 * (size_t)cap);
        info = c.st.scratch.get<double>("orth_info", 2 * (size_t)cap);
        SCANRS_HIP(hipMemsetAsync(ctl, 0, 2 * (size_t)cap * sizeof(int), c.s));
        SCANRS_HIP(hipMemsetAsync(info, 0, 2 * (size_t)cap * sizeof(double), c.s));
        used = 0;
    }
    uint32_t next() {
        if (used >= cap) fail(SCANRS_ERR_NUMERICAL, "orthonormalisation bookkeeping exhausted (%u slots)", cap);
        return used++;
    }
};
constexpr int DEV_CHOLQR_PASSES = 3; // two applications and a final check (the host loop usually stops at its second Gram matrix)

// coef (coef_rows x n, ld ldcoef, device; optional) follows P: every right-multiplication applied to P is applied to it
static void orth_cholqr_dev(Ctx &c, DevOrth &od, double *P, double *tmp, uint32_t ld, uint32_t n, uint64_t rows, double *coef = nullptr,
                            double *coef_tmp = nullptr, uint32_t coef_rows = 0, uint32_t ldcoef = 0) {
    const uint32_t slot = od.next();
    int *ctl = od.ctl + 2 * slot;
    double *info = od.info + 2 * slot;
    double *dG = c.dev(c.st.skey("orth_G").c_str(), (size_t)n * n);
    double *dR = c.dev(c.st.skey("orth_Rinv").c_str(), (size_t)n * n);
    struct SkipScope { // the dense kernels queued inside return at once when this orthonormalisation has converged
        Storage &st;
        ~SkipScope() { st.skip_flag = nullptr; }
    } scope{c.st};
    for (int pass = 0; pass < DEV_CHOLQR_PASSES; pass++) {
        const bool last = pass + 1 == DEV_CHOLQR_PASSES;
        if (pass >= 1) c.st.skip_flag = ctl; // ctl[0]: done. The Gram kernel of pass 1 still runs (nothing has converged before its check)
        launch_gram(c.st, P, ld, n, P, ld, n, rows, dG);
        launch_chol_rinv(c.st, dG, n, rows, pass, last, ctl, dR, info);
        if (last) break;
        launch_gemm_nn(c.st, P, ld, n, dR, n, n, rows, 1.0, 0.0, nullptr, 0, tmp, ld);
        launch_copy_cols(c.st, tmp, ld, P, ld, rows, ld);
        if (coef) {
            launch_gemm_nn(c.st, coef, ldcoef, n, dR, n, n, coef_rows, 1.0, 0.0, nullptr, 0, coef_tmp, ldcoef);
            launch_copy_cols(c.st, coef_tmp, ldcoef, coef, ldcoef, coef_rows, ldcoef);
        }
    }
}

// Block Bj against the first nprev columns of Q and within itself: two rounds of (project, CholeskyQR) — "twice is enough";
// the host loop's third Gram matrix, which only confirms that the projection has reached rounding level, is the final check here.
// coef (q x b, ld ldb) / cfull (q x q, ld ldcf): Bj = Korig coef, Q = Korig cfull, both on the device.
static void orth_against_dev(Ctx &c, DevOrth &od, const double *Q, uint32_t ldq, uint32_t nprev, double *Bj, double *tmp, uint32_t ldb,
                             uint32_t b, uint64_t rows, double *coef, double *coef_tmp, const double *cfull, uint32_t q, uint32_t ldcf) {
    double *dC = nprev ? c.dev(c.st.skey("orth_C").c_str(), (size_t)nprev * b) : nullptr;
    for (int round = 0; round < 2; round++) {
        if (nprev) {
            launch_gram(c.st, Q, ldq, nprev, Bj, ldb, b, rows, dC);
            launch_gemm_nn(c.st, Q, ldq, nprev, dC, b, b, rows, -1.0, 1.0, Bj, ldb, Bj, ldb);
            // coef -= cfull[:, :nprev] C; rows >= nprev of those columns of cfull are zero (block upper triangular)
            launch_gemm_nn(c.st, cfull, ldcf, nprev, dC, b, b, nprev, -1.0, 1.0, coef, ldb, coef, ldb);
        }
        orth_cholqr_dev(c, od, Bj, tmp, ldb, b, rows, coef, coef_tmp, q, ldb);
        if (!nprev) return;
    }
    const uint32_t slot = od.next();
    launch_gram(c.st, Q, ldq, nprev, Bj, ldb, b, rows, dC);
    launch_absmax_flag(c.st, dC, nprev * b, 1e-14, od.ctl + 2 * slot, od.info + 2 * slot);
}

// Which side of the view is long / sharded
static bool dim_sharded_rows(const scanrs_mat *m) { return rows_sharded(m); }
static bool dim_sharded_cols(const scanrs_mat *m) { return cols_sharded(m); }

// One-sided Jacobi (Hestenes) SVD of a tall host matrix A (rows x n, row-major): A = U diag(S) V^T with U rows x n.
// Used where the Gram-matrix route of ritz_finish would lose the small singular values (it squares the condition
// number): requested components at or beyond the numerical rank. Columns with S = 0 get U = 0.
static void jacobi_svd_tall(std::vector<double> A, size_t rows, int n, std::vector<double> &U, std::vector<double> &S,
                            std::vector<double> &V) {
    // work column-major for contiguous column sweeps
    std::vector<double> W((size_t)n * rows);
    for (size_t i = 0; i < rows; i++)
        for (int j = 0; j < n; j++) W[(size_t)j * rows + i] = A[i * n + j];
    std::vector<double> Vc((size_t)n * n, 0.0); // column-major: Vc[j * n + i] = V[i][j]
    for (int i = 0; i < n; i++) Vc[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 80; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double *__restrict__ wp = &W[(size_t)p * rows], *__restrict__ wq = &W[(size_t)q * rows];
                double a = 0, b = 0, g = 0;
                for (size_t i = 0; i < rows; i++) {
                    a += wp[i] * wp[i];
                    b += wq[i] * wq[i];
                    g += wp[i] * wq[i];
                }
                if (g == 0.0 || !(a > 0.0) || !(b > 0.0)) continue;
                const double rel = std::fabs(g) / std::sqrt(a * b);
                off = std::max(off, rel);
                if (rel <= 1e-16) continue;
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (size_t i = 0; i < rows; i++) {
                    const double x = wp[i], y = wq[i];
                    wp[i] = cs * x - sn * y;
                    wq[i] = sn * x + cs * y;
                }
                double *vp = &Vc[(size_t)p * n], *vq = &Vc[(size_t)q * n];
                for (int i = 0; i < n; i++) {
                    const double x = vp[i], y = vq[i];
                    vp[i] = cs * x - sn * y;
                    vq[i] = sn * x + cs * y;
                }
            }
        if (off < 1e-15) break;
    }
    std::vector<double> sv(n);
    std::vector<int> ord(n);
    for (int j = 0; j < n; j++) {
        double a = 0;
        for (size_t i = 0; i < rows; i++) a += W[(size_t)j * rows + i] * W[(size_t)j * rows + i];
        sv[j] = std::sqrt(a);
        ord[j] = j;
    }
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return sv[x] > sv[y]; });
    U.assign(rows * n, 0.0);
    V.assign((size_t)n * n, 0.0);
    S.assign(n, 0.0);
    for (int jj = 0; jj < n; jj++) {
        const int j = ord[jj];
        S[jj] = sv[j];
        if (sv[j] > 0)
            for (size_t i = 0; i < rows; i++) U[i * n + jj] = W[(size_t)j * rows + i] / sv[j];
        for (int i = 0; i < n; i++) V[(size_t)i * n + jj] = Vc[(size_t)j * n + i];
    }
}

// Rayleigh-Ritz when requested components reach the numerical rank of the projection T and T is SHARDED (or too large to take to
// the host): the reference's svddc on T (bk_svd.rs:105,134; rand_svd.rs:96,118) still returns accurate small singular values
// and a complete orthonormal set, and so must this (round 5 refused with SCANRS_ERR_NUMERICAL). Everything is decided from
// all-reduced Gram matrices, which every rank holds bit for bit - all ranks take the same branches and meet in the same
// collectives - in LEVELS: the eigenvectors of G = T^T T with w_j > 1e-12 w_1 are taken as they are (the accuracy the regular path
// has); the rest of the spectrum is looked at again through T B, B = the complement's basis, recomputed from T itself so that
// its Gram matrix is accurate at ITS scale; directions whose eigenvalue lies below the rounding floor of T (1e-28 w_1) are exact
// zeros: sigma = 0, the S-side vectors any orthonormal completion inside the complement, the T-side vectors a fixed pseudo-random
// function of (global row, column) made orthogonal to the columns found (two projections) and to each other (CholeskyQR). One
// CholeskyQR over all k T-side columns polishes what sigma_1 / sigma_j lost.
static void ritz_deficient_by_levels(Ctx &c, const double *Q, uint32_t ldq, uint32_t q, uint64_t ds, const double *T, uint32_t ldt, uint64_t dt, bool t_sharded,
                                     uint32_t k, const std::vector<double> &G0, double *dS, double *dT, uint32_t ldk, double *hSigma) {
    Tick tk("ritz: rank-deficient projection on a sharded side, by levels of the reduced Gram matrix");
    std::vector<double> B((size_t)q * q, 0.0); // q x m, row-major: the current complement's orthonormal basis (level 0: the identity)
    for (uint32_t i = 0; i < q; i++) B[(size_t)i * q + i] = 1.0;
    uint32_t m = q, done = 0;
    std::vector<double> Gm = G0, E((size_t)q * k, 0.0), Es((size_t)q * k, 0.0);
    double w_first = 0.0;
    std::vector<double> w, Z;
    for (int level = 0; level < 4 && done < k && m > 0; level++) {
        w.assign(m, 0.0);
        Z.assign((size_t)m * m, 0.0);
        for (uint32_t i = 0; i < m; i++)
            for (uint32_t j = i + 1; j < m; j++) {
                const double a = 0.5 * (Gm[(size_t)i * m + j] + Gm[(size_t)j * m + i]);
                if (!std::isfinite(a)) fail(SCANRS_ERR_NUMERICAL, "Rayleigh-Ritz: non-finite Gram matrix");
                Gm[(size_t)i * m + j] = Gm[(size_t)j * m + i] = a;
            }
        if (!sym_eig(Gm.data(), (int)m, w.data(), Z.data())) fail(SCANRS_ERR_NUMERICAL, "eigensolver did not converge");
        if (level == 0) w_first = w[0];
        if (!(w[0] > 1e-28 * w_first) || !(w_first > 0.0)) break; // what is left is rounding noise of T: exact zeros from here on
        uint32_t r = 0;
        while (r < m && w[r] > 1e-12 * w[0] && w[r] > 1e-28 * w_first) r++;
        const uint32_t take = std::min(k - done, r);
        for (uint32_t j = 0; j < take; j++) { // E[:, done + j] = B Z[:, j]
            const double sig = std::sqrt(w[j]);
            hSigma[done + j] = sig;
            for (uint32_t i = 0; i < q; i++) {
                double a = 0.0;
                for (uint32_t t = 0; t < m; t++) a += B[(size_t)i * m + t] * Z[(size_t)t * m + j];
                E[(size_t)i * k + done + j] = a;
                Es[(size_t)i * k + done + j] = a / sig;
            }
        }
        done += take;
        if (done == k || r == m) break;
        // the complement of the directions this level resolved: B <- B Z[:, r:], its projection recomputed from T
        const uint32_t m2 = m - r;
        std::vector<double> B2((size_t)q * m2);
        for (uint32_t i = 0; i < q; i++)
            for (uint32_t j = 0; j < m2; j++) {
                double a = 0.0;
                for (uint32_t t = 0; t < m; t++) a += B[(size_t)i * m + t] * Z[(size_t)t * m + r + j];
                B2[(size_t)i * m2 + j] = a;
            }
        B.swap(B2);
        m = m2;
        const uint32_t ldm = even_up(m);
        double *T2 = c.dev("ritz_lvl_t", (size_t)dt * ldm);
        gemm_hostw(c, T, ldt, q, B, m, dt, 1.0, 0.0, T2, ldm, "ritz_lvl_b");
        gram_host(c, T2, ldm, m, T2, ldm, m, dt, t_sharded, Gm);
        w.clear();
    }
    const uint32_t nz = k - done; // exact zeros
    if (nz) {
        // S side: any orthonormal vectors of the complement that were not taken. If the last level was decomposed (w non-empty) its
        // eigenvectors beyond the taken ones, else the complement's basis itself.
        for (uint32_t j = 0; j < nz; j++) {
            hSigma[done + j] = 0.0;
            for (uint32_t i = 0; i < q; i++) {
                double a = 0.0;
                if (!w.empty()) {
                    const uint32_t col = std::min<uint32_t>(m - 1u, (uint32_t)(m - nz + j)); // the smallest eigenvalues' directions: never among the taken ones (taken + nz <= m)
                    for (uint32_t t = 0; t < m; t++) a += B[(size_t)i * m + t] * Z[(size_t)t * m + col];
                } else {
                    a = B[(size_t)i * m + std::min<uint32_t>(m - 1u, j)];
                }
                E[(size_t)i * k + done + j] = a;
                Es[(size_t)i * k + done + j] = 0.0;
            }
        }
    }
    double *dE = c.dev("ritz_e", (size_t)q * k);
    c.h2d(dE, E.data(), E.size());
    launch_gemm_nn(c.st, Q, ldq, q, dE, k, k, ds, 1.0, 0.0, nullptr, 0, dS, ldk);
    double *dEs = c.dev("ritz_es", (size_t)q * k);
    c.h2d(dEs, Es.data(), Es.size());
    launch_gemm_nn(c.st, T, ldt, q, dEs, k, k, dt, 1.0, 0.0, nullptr, 0, dT, ldk); // (the zero columns of Es leave zero columns)
    if (nz) {
        const uint64_t row0 = t_sharded ? c.st.shard.outer_begin : 0;
        const uint32_t ldr = even_up(nz);
        double *R = c.dev("ritz_lvl_r", (size_t)dt * ldr), *Rtmp = c.dev("ritz_lvl_rtmp", (size_t)dt * ldr);
        launch_fill_hash(c.st, R, ldr, dt, row0, 0, nz, 0x71a2d3c5ull + k);
        for (int pass = 0; pass < 2 && done; pass++) { // R -= U1 (U1^T R), U1 = the columns found
            std::vector<double> C;
            gram_host(c, dT, ldk, done, R, ldr, nz, dt, t_sharded, C);
            gemm_hostw(c, dT, ldk, done, C, nz, dt, -1.0, 1.0, R, ldr, "ritz_lvl_c");
        }
        orth_cholqr(c, R, Rtmp, ldr, nz, dt, t_sharded, nullptr, 0, false);
        launch_copy_cols(c.st, R, ldr, dT + done, ldk, dt, nz);
    }
    double *ptmp = c.dev("ritz_lvl_ptmp", (size_t)dt * ldk);
    orth_cholqr(c, dT, ptmp, ldk, k, dt, t_sharded, nullptr, 0, false); // polish
}

// Rayleigh-Ritz finish shared by svd_bk and svd_rand: given an orthonormal Q on side S (dimension ds,
// q columns), T = op(Q) on the other side (dimension dt), return the top-k triplets.
//   side_S_vectors = Q * E, side_T_vectors = T * E * Sigma^-1, sigma = sqrt(eig(T^T T)).
static void ritz_finish(Ctx &c, const double *Q, uint32_t ldq, uint32_t q, uint64_t ds, const double *T, uint32_t ldt,
                        uint64_t dt, bool t_sharded, uint32_t k, double *hS, double *hSigma, double *hT) {
    std::vector<double> G;
    stage_mark("ritz gram");
    gram_host(c, T, ldt, q, T, ldt, q, dt, t_sharded, G);
    stage_mark("ritz gram done");
    for (uint32_t i = 0; i < q; i++) // symmetrise against rounding asymmetry
        for (uint32_t j = i + 1; j < q; j++) {
            const double a = 0.5 * (G[(size_t)i * q + j] + G[(size_t)j * q + i]);
            G[(size_t)i * q + j] = G[(size_t)j * q + i] = a;
        }
    std::vector<double> w(k), Z((size_t)q * k);
    {
        Tick tk("ritz: host sym_eig_topk");
        if (!sym_eig_topk(G.data(), (int)q, (int)k, w.data(), Z.data()))
            fail(SCANRS_ERR_NUMERICAL, "eigensolver did not converge");
    }
    stage_mark("ritz eig done");
    const uint32_t ldk = even_up(k);
    double *dS = c.dev("ritz_s", (size_t)ds * ldk);
    double *dT = c.dev("ritz_t", (size_t)dt * ldk);
    // sigma_i^2 are eigenvalues of T^T T: relative accuracy eps (sigma_1 / sigma_i)^2. When the k-th requested value sits
    // at or below the numerical rank (w_k <= 1e-12 w_1: k near min(m, n), rank-deficient input) the reference's svddc on T
    // is still accurate and orthonormal; the Gram route is not. Small problems go through a one-sided Jacobi SVD of T on
    // the host; large ones are refused rather than answered with zero or inaccurate vectors.
    if (!(w[k - 1] > 1e-12 * w[0]) || !(w[0] > 0.0)) {
        if (t_sharded || (double)dt * q > 3.0e7) {
            ritz_deficient_by_levels(c, Q, ldq, q, ds, T, ldt, dt, t_sharded, k, G, dS, dT, ldk, hSigma);
            if (hS) download_panel(c, dS, ldk, ds, k, hS);
            if (hT) download_panel(c, dT, ldk, dt, k, hT);
            c.sync();
            c.st.pca_dev.k = k;
            c.st.pca_dev.ld_u = c.st.pca_dev.ld_v = ldk;
            c.st.pca_dev.u = dS;
            c.st.pca_dev.rows_u = ds;
            c.st.pca_dev.v = dT;
            c.st.pca_dev.rows_v = dt;
            return;
        }
        Tick tk("ritz: host Jacobi SVD of T (rank-deficient)");
        std::vector<double> hTm((size_t)dt * q), Uj, Sj, Vj;
        download_panel(c, T, ldt, dt, q, hTm.data());
        jacobi_svd_tall(std::move(hTm), (size_t)dt, (int)q, Uj, Sj, Vj);
        std::vector<double> Ek((size_t)q * k), Uk((size_t)dt * k);
        for (uint32_t j = 0; j < k; j++) {
            hSigma[j] = Sj[j];
            for (uint32_t i = 0; i < q; i++) Ek[(size_t)i * k + j] = Vj[(size_t)i * q + j];
            for (uint64_t i = 0; i < dt; i++) Uk[i * k + j] = Uj[i * q + j];
        }
        // columns of the T side beyond the rank: any orthonormal completion (the reference's LAPACK returns one); here by
        // Gram-Schmidt of unit vectors against the columns already present
        for (uint32_t j = 0; j < k; j++) {
            if (Sj[j] > 0.0) continue;
            for (uint64_t e0 = 0; e0 < dt; e0++) {
                std::vector<double> cand(dt, 0.0);
                cand[e0] = 1.0;
                for (int pass = 0; pass < 2; pass++)
                    for (uint32_t jj = 0; jj < k; jj++) {
                        if (jj == j || (jj > j && !(Sj[jj] > 0.0))) continue;
                        double d = 0;
                        for (uint64_t i = 0; i < dt; i++) d += cand[i] * Uk[i * k + jj];
                        for (uint64_t i = 0; i < dt; i++) cand[i] -= d * Uk[i * k + jj];
                    }
                double nn = 0;
                for (uint64_t i = 0; i < dt; i++) nn += cand[i] * cand[i];
                if (nn > 0.25) {
                    nn = std::sqrt(nn);
                    for (uint64_t i = 0; i < dt; i++) Uk[i * k + j] = cand[i] / nn;
                    break;
                }
            }
        }
        double *dE2 = c.dev("ritz_e", (size_t)q * k);
        c.h2d(dE2, Ek.data(), Ek.size());
        launch_gemm_nn(c.st, Q, ldq, q, dE2, k, k, ds, 1.0, 0.0, nullptr, 0, dS, ldk);
        upload_panel(c, Uk.data(), dt, k, dT, ldk);
        if (hS) download_panel(c, dS, ldk, ds, k, hS);
        if (hT) memcpy(hT, Uk.data(), Uk.size() * 8);
        c.sync();
        c.st.pca_dev.k = k;
        c.st.pca_dev.ld_u = c.st.pca_dev.ld_v = ldk;
        c.st.pca_dev.u = dS;
        c.st.pca_dev.rows_u = ds;
        c.st.pca_dev.v = dT;
        c.st.pca_dev.rows_v = dt;
        return;
    }
    std::vector<double> E((size_t)q * k), Es((size_t)q * k);
    for (uint32_t j = 0; j < k; j++) {
        const double sig = std::sqrt(std::max(w[j], 0.0));
        hSigma[j] = sig;
        const double inv = sig > 0.0 ? 1.0 / sig : 0.0;
        for (uint32_t i = 0; i < q; i++) {
            E[(size_t)i * k + j] = Z[(size_t)i * k + j];
            Es[(size_t)i * k + j] = Z[(size_t)i * k + j] * inv;
        }
    }
    double *dE = c.dev("ritz_e", (size_t)q * k);
    c.h2d(dE, E.data(), E.size());
    launch_gemm_nn(c.st, Q, ldq, q, dE, k, k, ds, 1.0, 0.0, nullptr, 0, dS, ldk);
    hipEvent_t ev_s = nullptr; // the S-side factor is in place
    SCANRS_HIP(hipEventCreateWithFlags(&ev_s, hipEventDisableTiming));
    struct EvS {
        hipEvent_t e;
        ~EvS() { (void)hipEventDestroy(e); }
    } ev_s_guard{ev_s};
    SCANRS_HIP(hipEventRecord(ev_s, c.s));
    double *dEs = c.dev("ritz_es", (size_t)q * k);
    c.h2d(dEs, Es.data(), Es.size());
    // The T-side factor (10^6 x 50 at the headline size: a 2.5 ms GEMM, then 400 MB to the host) in row blocks: the copy of a finished block
    // runs on a second stream beside the GEMM of the later ones (round 5; before: GEMM, then the whole copy)
    const bool piped = hT != nullptr && dt * (uint64_t)k * 8 >= (64u << 20) && ldk == k;
    if (!piped) {
        launch_gemm_nn(c.st, T, ldt, q, dEs, k, k, dt, 1.0, 0.0, nullptr, 0, dT, ldk);
        stage_mark("ritz factors sync");
        c.sync();
        stage_mark("ritz factors synced");
        Tick tk("ritz: download of the factors");
        const auto t0 = std::chrono::steady_clock::now();
        if (hS) download_panel(c, dS, ldk, ds, k, hS); // null: the caller keeps the result in HBM (scanrs_pca_result_device)
        if (hT) download_panel(c, dT, ldk, dt, k, hT);
        c.sync();
        c.st.t_delivery_us += (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    } else {
        constexpr int NBLK = 8;
        const uint64_t blk = (((dt + NBLK - 1) / NBLK) + 1023) & ~1023ull;
        hipEvent_t bev[NBLK];
        int n_blk = 0;
        for (uint64_t r0 = 0; r0 < dt; r0 += blk, n_blk++) {
            const uint64_t nr = std::min<uint64_t>(blk, dt - r0);
            launch_gemm_nn(c.st, T + r0 * ldt, ldt, q, dEs, k, k, nr, 1.0, 0.0, nullptr, 0, dT + r0 * ldk, ldk);
            SCANRS_HIP(hipEventCreateWithFlags(&bev[n_blk], hipEventDisableTiming));
            SCANRS_HIP(hipEventRecord(bev[n_blk], c.s));
        }
        Tick tk("ritz: download of the factors (beside the last GEMM)");
        const auto t0 = std::chrono::steady_clock::now();
        hipStream_t cs = c.st.aux2();
        int waited = -1;
        try {
            if (hS) { // the short side first, on the copy stream: it travels while the GEMM blocks of the long side run
                SCANRS_HIP(hipStreamWaitEvent(cs, ev_s, 0));
                if (ds * (uint64_t)k * 8 >= (8u << 20))
                    download_panel_staged(c, dS, ldk, ds, k, hS, cs);
                else
                    SCANRS_D2H_2D(hS, dS, (size_t)ldk * 8, (size_t)k * 8, ds, cs);
            }
            download_panel_staged(c, dT, ldk, dt, k, hT, cs, [&](uint64_t r0, uint64_t r1) {
                const int need = (int)((r1 - 1) / blk);
                while (waited < need) SCANRS_HIP(hipStreamWaitEvent(cs, bev[++waited], 0));
            });
            c.sync();
        } catch (...) {
            for (int i = 0; i < n_blk; i++) (void)hipEventDestroy(bev[i]);
            throw;
        }
        for (int i = 0; i < n_blk; i++) (void)hipEventDestroy(bev[i]);
        c.st.t_delivery_us += (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    // where the factors live on the device: side S first, side T second (the drivers map them to U / V)
    c.st.pca_dev.k = k;
    c.st.pca_dev.ld_u = c.st.pca_dev.ld_v = ldk;
    c.st.pca_dev.u = dS;
    c.st.pca_dev.rows_u = ds;
    c.st.pca_dev.v = dT;
    c.st.pca_dev.rows_v = dt;
}
// ritz_finish leaves (side S, side T) in pca_dev.(u, v); swap when side S holds V
static void pca_dev_swap(Storage &st) {
    std::swap(st.pca_dev.u, st.pca_dev.v);
    std::swap(st.pca_dev.rows_u, st.pca_dev.rows_v);
}

constexpr int BK_RETRY_ON_HOST = -1000; // device-side factorizations did not converge: run again with host factorizations
static int pca_bk_impl(scanrs_mat *m, uint32_t k, double k_multiplier, uint32_t n_iter, uint64_t seed, const double *omega,
                       const scanrs_snoop *snoop, double *u, double *s, double *v, bool device_factor) {
    Tick tk_all("bk: total");
    Ctx c(m);
    stage_mark("bk setup");
    // work is queued far ahead of the device here: whatever ends this call early (cancellation, a numerical failure) waits for
    // both streams before the scratch buffers change hands
    struct Drain {
        Storage &st;
        int n0 = std::uncaught_exceptions();
        ~Drain() {
            if (std::uncaught_exceptions() > n0) {
                try {
                    st.side_join_if(nullptr, true); // the helper thread that builds the second orientation, if it still runs
                } catch (const Failure &) {
                }
                (void)wait_stream_quiet(st.stream);
                if (st.aux_stream) (void)wait_stream_quiet(st.aux_stream);
                if (st.aux2_stream) (void)wait_stream_quiet(st.aux2_stream);
                if (st.ov_stream) (void)wait_stream_quiet(st.ov_stream);
            }
        }
    } drain{c.st};
    const uint64_t M = m->rows(), N = m->cols();
    // global extents decide the branch and the validation, as the reference sees the whole matrix
    const uint64_t Mg = dim_sharded_rows(m) ? c.st.shard.outer_global : M;
    const uint64_t Ng = dim_sharded_cols(m) ? c.st.shard.outer_global : N;
    if (Mg < 2 || Ng < 2) fail(SCANRS_ERR_SHAPE, "The input matrix must be at least 2x2.");
    if (k > std::min(Mg, Ng)) fail(SCANRS_ERR_INVALID_K, "invalid k");
    if (k == 0 || n_iter == 0) fail(SCANRS_ERR_ARGUMENT, "k and n_iter must be positive");
    uint32_t b = (uint32_t)std::ceil((double)k * k_multiplier); // bk_svd.rs:49
    b = (uint32_t)std::min<uint64_t>(std::min(Mg, Ng), b);     // bk_svd.rs:81
    if (b < k) fail(SCANRS_ERR_INVALID_K, "invalid k");

    const bool rows_ge = Mg >= Ng; // bk_svd.rs:89 `if m >= n`
    // S = the side the Krylov panel lives on (the shorter one): cols when m >= n, rows otherwise.
    const uint64_t ds = rows_ge ? N : M, dt = rows_ge ? M : N;
    const bool s_sharded = rows_ge ? dim_sharded_cols(m) : dim_sharded_rows(m);
    const bool t_sharded = rows_ge ? dim_sharded_rows(m) : dim_sharded_cols(m);
    if (s_sharded) fail(SCANRS_ERR_ARGUMENT, "shard the longer dimension of the matrix, not the shorter one");
    // op S->T: A*X when m >= n (X on the cols side), A^T*X otherwise
    const bool to_t_transpose = !rows_ge;

    if ((uint64_t)b * n_iter > ds) {
        // The Krylov matrix K (ds x b n_iter) is wider than it is tall: `K.qr()` (bk_svd.rs:98,127) then returns a square Q
        // spanning the whole side, T = Q^T A loses nothing and the driver returns the exact truncated SVD (small feature
        // panels: antibody / targeted-gene matrices, k close to min(m, n)). Any orthonormal basis of the whole space gives
        // the same singular triplets, so Q = I: one sparse product with the identity panel, then the Rayleigh-Ritz finish.
        const uint32_t qe = (uint32_t)ds, ldqe = even_up(qe);
        double *Ke = c.dev("bk_K", (size_t)ds * ldqe);
        double *Te = c.dev("bk_T", (size_t)dt * ldqe);
        std::vector<double> eye((size_t)ds * qe, 0.0);
        for (uint32_t i = 0; i < qe; i++) eye[(size_t)i * qe + i] = 1.0;
        SCANRS_HIP(hipMemsetAsync(Ke, 0, (size_t)ds * ldqe * 8, c.s));
        upload_panel(c, eye.data(), ds, qe, Ke, ldqe);
        for (uint32_t i = 0; i < n_iter; i++) progress_check(snoop, (double)i / (double)n_iter * 0.8); // the reference's poll points
        mat_apply(m, to_t_transpose, Ke, ldqe, qe, Te, ldqe);
        c.sync();
        progress_check(snoop, 0.93);
        if (rows_ge) {
            ritz_finish(c, Ke, ldqe, qe, ds, Te, ldqe, dt, t_sharded, k, v, s, u);
            pca_dev_swap(c.st);
        } else {
            ritz_finish(c, Ke, ldqe, qe, ds, Te, ldqe, dt, t_sharded, k, u, s, v);
        }
        progress_check(snoop, 1.0);
        return SCANRS_OK;
    }
    // the copy / tile layout of the SECOND product on a helper thread, beside everything up to the end of the first pass
    prepare_second_orientation(m, !to_t_transpose);
    const uint32_t ldb = even_up(b), q = b * n_iter, ldq = even_up(q);
    double *P = c.dev("bk_P", (size_t)ds * ldb);
    double *Ptmp = c.dev("bk_Ptmp", (size_t)ds * ldb);
    double *Y = c.dev("bk_Y", (size_t)dt * ldb);
    double *K = c.dev("bk_K", (size_t)ds * ldq);
    SCANRS_HIP(hipMemsetAsync(P, 0, (size_t)ds * ldb * 8, c.s));
    SCANRS_HIP(hipMemsetAsync(K, 0, (size_t)ds * ldq * 8, c.s));

    { // start panel: m >= n -> (n x b) as is; n > m -> reference holds (b x m), we hold its transpose
        Tick tk("bk: start panel");
        struct Acc {
            uint64_t &t;
            std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
            ~Acc() { t += (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
        } acc{c.st.t_start_panel_us};
        if (!omega) {
            // reference order: (n x b) row-major when m >= n, else (b x m) row-major held transposed here
            if (rows_ge)
                omega_fill_device(c.st, seed, ds, b, P, ldb, false);
            else
                omega_fill_device(c.st, seed, b, ds, P, ldb, true);
        } else if (rows_ge) {
            upload_panel(c, omega, ds, b, P, ldb);
        } else { // (b x ds) row-major on the host side: transposed on the device
            double *tmp = c.dev("bk_omega_t", (size_t)b * ds);
            c.h2d(tmp, omega, (size_t)b * ds);
            launch_transpose(c.st, tmp, b, ds, P, ldb);
        }
    }

    // The projection T = Q^T A (bk_svd.rs:104,131) is not recomputed as one q-wide sparse product: with
    // K = [K_0 .. K_{n-1}] the blocks A^T K_i (resp. A K_i) for i < n-1 are exactly the first half-products of
    // iterations 1..n-1, so they are written straight into T; one extra b-wide product supplies the last block
    // and T' = (op(A) K) C with Q = K C (C small, tracked on the host through the orthonormalisation).
    const bool reuse = (b % 2u) == 0u;
    const uint64_t fallbacks0 = c.st.orth_fallbacks; // a panel completed with random directions voids Q = K C (orth_host_mgs)
    double *T = c.dev("bk_T", (size_t)dt * ldq);
    // Q = qr(K).Q: block i is already orthonormal; orthogonalise it against blocks < i (in block order), tracking
    // Q = K C in `cfull`
    std::vector<double> cfull((size_t)q * q, 0.0);
    for (uint32_t i = 0; i < q; i++) cfull[(size_t)i * q + i] = 1.0;
    double *Bj = c.dev("bk_Bj", (size_t)ds * ldb);
    double *Btmp = c.dev("bk_Btmp", (size_t)ds * ldb);
    uint32_t next_block = 1;
    // device_factor: the factors and the coefficient matrix stay on the device (orth_*_dev above); the host copy of cfull is
    // filled once, when the basis is complete
    const bool dv = device_factor && chol_rinv_ok(b);
    DevOrth od;
    double *cfull_d = nullptr, *coef_d = nullptr, *coef_tmp_d = nullptr;
    if (dv) {
        od.init(c, 4u * n_iter + 8u);
        cfull_d = c.dev("bk_cfull", (size_t)q * ldq);
        coef_d = c.dev("bk_coef", (size_t)q * ldb);
        coef_tmp_d = c.dev("bk_coef_tmp", (size_t)q * ldb);
        SCANRS_HIP(hipMemcpy2DAsync(cfull_d, (size_t)ldq * 8, cfull.data(), (size_t)q * 8, (size_t)q * 8, q, hipMemcpyHostToDevice, c.s));
        c.sync(); // cfull is pageable host memory
    }
    auto orth_block_dev = [&](Ctx &cx, uint32_t i) {
        launch_copy_cols(cx.st, K + (size_t)i * b, ldq, Bj, ldb, ds, b);
        launch_copy_cols(cx.st, cfull_d + (size_t)i * b, ldq, coef_d, ldb, q, b); // the identity block
        orth_against_dev(cx, od, K, ldq, i * b, Bj, Btmp, ldb, b, ds, coef_d, coef_tmp_d, cfull_d, q, ldq);
        launch_copy_cols(cx.st, Bj, ldb, K + (size_t)i * b, ldq, ds, b);
        launch_copy_cols(cx.st, coef_d, ldb, cfull_d + (size_t)i * b, ldq, q, b);
        next_block = i + 1;
    };
    auto orth_block_on = [&](Ctx &cx, uint32_t i) {
        std::vector<double> coef((size_t)q * b, 0.0);
        launch_copy_cols(cx.st, K + (size_t)i * b, ldq, Bj, ldb, ds, b);
        for (uint32_t j = 0; j < b; j++) coef[(size_t)(i * b + j) * b + j] = 1.0;
        orth_against(cx, K, ldq, i * b, Bj, Btmp, ldb, b, ds, false, &coef, &cfull, q);
        launch_copy_cols(cx.st, Bj, ldb, K + (size_t)i * b, ldq, ds, b);
        for (uint32_t r = 0; r < q; r++)
            for (uint32_t j = 0; j < b; j++) cfull[(size_t)r * q + i * b + j] = coef[(size_t)r * b + j];
        next_block = i + 1;
    };
    auto orth_block = [&](uint32_t i) { // on the auxiliary stream, finished (host-synchronised) on return
        StreamSwap sw(c.st, c.st.aux());
        Ctx cx(m);
        orth_block_on(cx, i);
        cx.sync();
    };
    // T'_j = (op(A) K)[:, 0:(j+1)b] * C[0:(j+1)b, block j] for the blocks whose coefficients are known before the end: a
    // dense MFMA GEMM queued on the auxiliary stream behind the pass that delivers the last half-product it reads, so it
    // runs in the gaps the sparse passes leave. Out of place: later blocks still need the half-products in T, so the projected
    // blocks are collected in a second panel T2 of the same shape (block 0, whose projection is the half-product itself, is
    // copied there by the first early projection), and the Rayleigh-Ritz step reads T2 — nothing is copied back at the end.
    // The projections have a stream of their own: on the stream of the Gram-Schmidt chains a 3 ms GEMM stood between the main
    // stream and the chain it was waiting for at the end of the iterations.
    const uint32_t n_early = (reuse && n_iter >= 3) ? n_iter - 2 : 0; // blocks 1 .. n_iter-2
    double *T2 = reuse ? c.dev("bk_T2", (size_t)dt * ldq) : nullptr;
    bool block0_copied = false;
    struct Ev {
        hipEvent_t e = nullptr;
        Ev() { SCANRS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
        ~Ev() { (void)hipEventDestroy(e); }
    };
    // one event object per iteration and purpose: an event is never recorded again while a wait on its previous record may still be queued
    std::vector<Ev> ev_k(n_iter);    // K block i is in place (main stream)
    std::vector<Ev> ev_pass(n_iter); // both passes of iteration i are done (main stream)
    std::vector<Ev> ev_aux(n_iter);  // the Gram-Schmidt chain queued in iteration i is done (auxiliary stream)
    std::vector<Ev> ev_proj(n_iter); // the projection queued in iteration i is done (projection stream)
    int last_aux = -1, last_proj = -1;
    std::vector<char> projected(n_iter, 0);
    auto project_block_early = [&](uint32_t j, uint32_t it) { // queued in iteration `it`: T block j is ready at ev_pass[it]
        const uint32_t nr = (j + 1) * b;
        std::vector<double> W;
        if (!dv) {
            W.resize((size_t)nr * b);
            for (uint32_t r = 0; r < nr; r++)
                for (uint32_t cc = 0; cc < b; cc++) W[(size_t)r * b + cc] = cfull[(size_t)r * q + j * b + cc];
        }
        StreamSwap sw(c.st, c.st.aux2());
        Ctx cx(m);
        SCANRS_HIP(hipStreamWaitEvent(cx.s, ev_pass[it].e, 0));
        if (!block0_copied) {
            launch_copy_cols(cx.st, T, ldq, T2, ldq, dt, b);
            block0_copied = true;
        }
        if (dv) { // the coefficients of block j: finished on the auxiliary stream (ev_aux was recorded behind them)
            SCANRS_HIP(hipStreamWaitEvent(cx.s, ev_aux[it].e, 0));
            launch_gemm_nn(cx.st, T, ldq, nr, cfull_d + (size_t)j * b, ldq, b, dt, 1.0, 0.0, nullptr, 0, T2 + (size_t)j * b, ldq);
        } else {
            char key[32];
            snprintf(key, sizeof(key), "bk_projw%u", j);
            double *dW = cx.dev(key, (size_t)nr * b);
            cx.h2d(dW, W.data(), W.size());
            launch_gemm_nn(cx.st, T, ldq, nr, dW, b, b, dt, 1.0, 0.0, nullptr, 0, T2 + (size_t)j * b, ldq);
        }
        SCANRS_HIP(hipEventRecord(ev_proj[it].e, cx.s));
        last_proj = (int)it;
        projected[j] = 1;
    };
    for (uint32_t i = 0; i < n_iter; i++) {
        Tick tk("bk: iteration");
        stage_mark("bk iteration", i);
        // m >= n: B = qr((A B)^T A)^T .Q  (bk_svd.rs:94);  n > m: T = (B A)^T; B = qr(A T).Q^T  (bk_svd.rs:122-123)
        double *Yi = (reuse && i >= 1) ? T + (size_t)(i - 1) * b : Y;
        const uint32_t ldy = (reuse && i >= 1) ? ldq : ldb;
        mat_apply(m, to_t_transpose, P, ldb, b, Yi, ldy);
        // Q = qr(K).Q block by block: block i-1 is complete since the end of the previous iteration, so it is made
        // orthogonal to blocks < i-1 now, on the auxiliary stream, while the sparse pass just queued occupies the
        // main one (host round trips of the small factorizations hidden behind ~40 ms of gather work).
        if (i >= 2) {
            if (dv) { // queued on the auxiliary stream, nothing waited for
                StreamSwap sw(c.st, c.st.aux());
                Ctx cx(m);
                SCANRS_HIP(hipStreamWaitEvent(cx.s, ev_k[i - 1].e, 0));
                orth_block_dev(cx, i - 1);
                SCANRS_HIP(hipEventRecord(ev_aux[i].e, cx.s));
                last_aux = (int)i;
            } else {
                orth_block(i - 1);
            }
        }
        mat_apply(m, !to_t_transpose, Yi, ldy, b, P, ldb);
        if (i >= 2 && n_early && i - 1 <= n_early) {
            // behind BOTH passes of this iteration: the projection GEMM then shares the device with the chain of small kernels that
            // orthonormalises the panel (and with the start of the next product, whose workgroups take their items dynamically)
            // instead of with the overflow gather's tail, which the product waits for
            SCANRS_HIP(hipEventRecord(ev_pass[i].e, c.s));
            project_block_early(i - 1, i);
        }
        if (trace_on()) { // separate the wait for the two passes from the factorization in the trace
            Tick tw("  passes (wait)");
            c.sync();
        }
        if (dv)
            orth_cholqr_dev(c, od, P, Ptmp, ldb, b, ds);
        else
            orth_cholqr(c, P, Ptmp, ldb, b, ds, false);
        launch_copy_cols(c.st, P, ldb, K + (size_t)i * b, ldq, ds, b);
        if (dv) {
            // the host stays one iteration ahead of the device: the queue never runs dry and a cancellation is seen within one iteration
            SCANRS_HIP(hipEventRecord(ev_k[i].e, c.s));
            if (trace_on())
                c.sync();
            else if (i >= 1) {
                stage_mark("bk pacing wait", i - 1);
                SCANRS_SYNC_EVENT(ev_k[i - 1].e);
                stage_mark("bk pacing done", i - 1);
            }
        } else {
            c.sync();
        }
        progress_check(snoop, (double)i / (double)n_iter * 0.8);
    }
    // the last block (and everything, when there was no iteration to hide behind) on the main stream
    {
        Tick tk("bk: orth(K)");
        if (dv) {
            if (last_aux >= 0) SCANRS_HIP(hipStreamWaitEvent(c.s, ev_aux[last_aux].e, 0)); // the blocks done beside the passes (one stream: the last record covers them all)
            for (uint32_t i = next_block; i < n_iter; i++) orth_block_dev(c, i);
            // the one look at what the device-side factorizations did, together with the coefficients
            std::vector<int> ctl(2 * (size_t)od.used);
            std::vector<double> info(2 * (size_t)od.used);
            stage_mark("bk verdict sync");
            SCANRS_D2H_2D(cfull.data(), cfull_d, (size_t)ldq * 8, (size_t)q * 8, q, c.s);
            SCANRS_D2H(ctl.data(), od.ctl, ctl.size() * sizeof(int), c.s);
            SCANRS_D2H(info.data(), od.info, info.size() * sizeof(double), c.s);
            stage_mark("bk verdict synced");
            for (uint32_t sl = 0; sl < od.used; sl++) {
                if (ctl[2 * sl + 1] == 2) fail(SCANRS_ERR_NUMERICAL, "orthonormalisation: non-finite Gram matrix");
                if (ctl[2 * sl] != 1 || ctl[2 * sl + 1] != 0) {
                    if (trace_on())
                        fprintf(stderr, "[scanrs trace] bk: device factorization %u: done %d status %d (last check %.3e, shift %.3e) -> host path\n", sl,
                                ctl[2 * sl], ctl[2 * sl + 1], info[2 * sl], info[2 * sl + 1]);
                    if (c.st.aux_stream) SCANRS_SYNC(c.st.aux_stream);
                    if (c.st.aux2_stream) SCANRS_SYNC(c.st.aux2_stream);
                    return BK_RETRY_ON_HOST;
                }
            }
        } else {
            for (uint32_t i = next_block; i < n_iter; i++) orth_block_on(c, i);
            c.sync();
        }
    }
    progress_check(snoop, 0.82);
    // T' = (op(A) K) C amplifies the rounding of op(A) K by |C|. Columns of Q whose coefficient column stays
    // small (<= 1e5: error <= ~1e-11) take the cheap dense route; the others — directions in which the Krylov
    // blocks are numerically dependent — are recomputed directly as a (narrow) sparse product op(A) Q[:, bad].
    double cmax_limit = c.st.reuse_cmax;
    std::vector<uint32_t> bad;
    const bool c_valid = c.st.orth_fallbacks == fallbacks0;
    if (reuse && !c_valid) {
        for (uint32_t j = 0; j < q; j++) bad.push_back(j); // everything directly: one q-wide product below
        if (trace_on()) fprintf(stderr, "[scanrs trace] bk: rank-deficient panels were completed on the host: the projection is computed directly\n");
    }
    if (reuse && c_valid) {
        std::vector<double> colmax(q, 0.0);
        for (uint32_t r = 0; r < q; r++)
            for (uint32_t j = 0; j < q; j++) colmax[j] = std::max(colmax[j], std::fabs(cfull[(size_t)r * q + j]));
        // The repair pass carries the last block (b columns) plus the flagged columns of the older blocks. A pass of up to 104
        // columns runs through the hybrid LDS-tile product, a wider one through the gather kernels at about twice the time: when
        // a few flagged columns too many stand in the way (22 on the 1 M-cell benchmark), the bound is raised — by at most two
        // decades, error <= ~1e-9 instead of 1e-11 on those columns, far inside every tolerance — until the pass fits.
        // The rule looks at b and the coefficients only — not at whether this handle (or this shard) happens to run the hybrid product:
        // the flagged set, and with it the rounding of the result, is the same on every rank of a sharded run and under any memory
        // pressure (it used to depend on mat_tiles_ready).
        if (n_iter >= 2 && b <= 104u) {
            const uint32_t last_lo = (n_iter - 1) * b, room = 104u - b;
            std::vector<double> flagged;
            for (uint32_t j = 0; j < last_lo; j++)
                if (!(colmax[j] < cmax_limit)) flagged.push_back(colmax[j]);
            if (flagged.size() > room) {
                std::sort(flagged.begin(), flagged.end(), std::greater<double>());
                const double need = flagged[room]; // the largest coefficient that has to take the dense route
                if (flagged[0] < 100.0 * cmax_limit) // all of them within the two decades: the pass carries the last block alone (100 columns run 9 % faster than 104)
                    cmax_limit = std::nextafter(flagged[0], INFINITY);
                else if (need < 100.0 * cmax_limit && (room == 0 || flagged[room - 1] > need))
                    cmax_limit = std::nextafter(need, INFINITY);
            }
        }
        for (uint32_t j = 0; j < q; j++)
            if (!(colmax[j] < cmax_limit)) bad.push_back(j);
        if (trace_on()) {
            fprintf(stderr, "[scanrs trace] bk: %zu of %u projection columns recomputed directly\n", bad.size(), q);
            for (uint32_t blk = 0; blk < n_iter; blk++) {
                int cnt = 0;
                for (uint32_t j = blk * b; j < (blk + 1) * b; j++) cnt += !(colmax[j] < cmax_limit);
                fprintf(stderr, "[scanrs trace] bk:   block %u: %d\n", blk, cnt);
            }
            for (double thr : {1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10}) {
                int cnt = 0;
                for (uint32_t j = 0; j < q; j++) cnt += !(colmax[j] < thr);
                fprintf(stderr, "[scanrs trace] bk:   columns with max|C| >= %.0e: %d\n", thr, cnt);
            }
        }
    }
    // The last Krylov block: its half-product op(A) K_{n-1} exists only to feed T'_{n-1}. When (as on every matrix
    // measured: the newest block is where the numerical dependence shows) most of that block is recomputed directly
    // anyway, computing ALL of it directly — in the same sparse pass as the other bad columns — costs no more
    // 128-column passes than the half-product plus the repair pass, and usually one fewer.
    bool last_direct = false;
    if (reuse && c_valid) {
        const uint32_t last_lo = (n_iter - 1) * b;
        uint32_t bad_other = 0;
        for (uint32_t j : bad) bad_other += j < last_lo;
        auto chunks = [](uint32_t x) { return (x + 127u) / 128u; };
        const uint32_t cost_half = chunks(b) + chunks((uint32_t)bad.size()), cost_direct = chunks(b + bad_other);
        last_direct = cost_direct <= cost_half && n_iter >= 2;
        if (last_direct) {
            std::vector<uint32_t> merged;
            for (uint32_t j : bad)
                if (j < last_lo) merged.push_back(j);
            for (uint32_t j = last_lo; j < q; j++) merged.push_back(j);
            bad.swap(merged);
        } else {
            Tick tk("bk: last block product");
            mat_apply(m, to_t_transpose, P, ldb, b, T + (size_t)(n_iter - 1) * b, ldq);
        }
        if (trace_on()) fprintf(stderr, "[scanrs trace] bk: last block %s\n", last_direct ? "computed directly with the repair pass" : "through its half-product");
    }
    double *Tfin = T; // the panel the Rayleigh-Ritz step reads
    {
        Tick tk("bk: projection");
        if (reuse && (bad.size() * 2 < q || last_direct)) {
            // T2 block j = T[:, 0:(j+1)b] * C[0:(j+1)b, block j] for the blocks not projected early
            std::vector<double> W;
            for (uint32_t j = n_iter - 1 - (last_direct ? 1u : 0u); j >= 1; j--) {
                if (projected[j]) continue; // done early
                const uint32_t nr = (j + 1) * b;
                if (dv) {
                    launch_gemm_nn(c.st, T, ldq, nr, cfull_d + (size_t)j * b, ldq, b, dt, 1.0, 0.0, nullptr, 0, T2 + (size_t)j * b, ldq);
                } else {
                    W.assign((size_t)nr * b, 0.0);
                    for (uint32_t r = 0; r < nr; r++)
                        for (uint32_t cc = 0; cc < b; cc++) W[(size_t)r * b + cc] = cfull[(size_t)r * q + j * b + cc];
                    double *dW = c.dev("bk_projw", (size_t)nr * b);
                    c.h2d(dW, W.data(), W.size());
                    launch_gemm_nn(c.st, T, ldq, nr, dW, b, b, dt, 1.0, 0.0, nullptr, 0, T2 + (size_t)j * b, ldq);
                }
            }
            if (!block0_copied) launch_copy_cols(c.st, T, ldq, T2, ldq, dt, b);
            bool any_early = false;
            for (uint32_t j = 1; j < n_iter; j++) any_early = any_early || projected[j];
            if (any_early && last_proj >= 0) SCANRS_HIP(hipStreamWaitEvent(c.s, ev_proj[last_proj].e, 0)); // the last early GEMM (they are ordered on their stream)
            Tfin = T2;
            if (!bad.empty()) {
                const uint32_t nbad = (uint32_t)bad.size(), ldbad = even_up(nbad);
                uint32_t *d_idx = c.st.scratch.get<uint32_t>("bk_badidx", nbad);
                SCANRS_HIP(hipMemcpyAsync(d_idx, bad.data(), (size_t)nbad * 4, hipMemcpyHostToDevice, c.s));
                c.sync();
                double *Qbad = c.dev("bk_Qbad", (size_t)ds * ldbad);
                double *Tbad = c.dev("bk_Tbad", (size_t)dt * ldbad);
                launch_permute_cols(c.st, K, ldq, Qbad, ldbad, ds, d_idx, nbad, false);
                mat_apply(m, to_t_transpose, Qbad, ldbad, nbad, Tbad, ldbad);
                launch_permute_cols(c.st, Tbad, ldbad, T2, ldq, dt, d_idx, nbad, true);
            }
        } else {
            bool any_early = false;
            for (uint32_t j = 1; j < n_iter; j++) any_early = any_early || projected[j];
            if (any_early && last_proj >= 0) SCANRS_HIP(hipStreamWaitEvent(c.s, ev_proj[last_proj].e, 0)); // they read T, which is overwritten now
            mat_apply(m, to_t_transpose, K, ldq, q, T, ldq);
        }
        stage_mark("bk projection sync");
        c.sync();
        stage_mark("bk projection synced");
    }
    Tick tk_fin("bk: ritz_finish");
    progress_check(snoop, 0.93);
    // m >= n: T = A Q (m x q): U = T E S^-1, V = Q E.   n > m: T^T = A^T Q (n x q): U = Q E, V = T E S^-1.
    if (rows_ge) {
        ritz_finish(c, K, ldq, q, ds, Tfin, ldq, dt, t_sharded, k, v, s, u);
        pca_dev_swap(c.st); // side S = cols: it holds V
    } else {
        ritz_finish(c, K, ldq, q, ds, Tfin, ldq, dt, t_sharded, k, u, s, v);
    }
    progress_check(snoop, 1.0);
    return SCANRS_OK;
}

int pca_bk(scanrs_mat *m, uint32_t k, double k_multiplier, uint32_t n_iter, uint64_t seed, const double *omega,
           const scanrs_snoop *snoop, double *u, double *s, double *v) {
    int rc = pca_bk_impl(m, k, k_multiplier, n_iter, seed, omega, snoop, u, s, v, m->st->device_factor != 0);
    if (rc == BK_RETRY_ON_HOST) {
        m->st->bk_host_retries++;
        rc = pca_bk_impl(m, k, k_multiplier, n_iter, seed, omega, snoop, u, s, v, false);
    }
    return rc;
}

int pca_rand(scanrs_mat *m, uint32_t k, double l_multiplier, uint32_t n_iter, uint64_t seed, const double *omega, double *u,
             double *s, double *v) {
    Ctx c(m);
    const uint64_t M = m->rows(), N = m->cols();
    const uint64_t Mg = dim_sharded_rows(m) ? c.st.shard.outer_global : M;
    const uint64_t Ng = dim_sharded_cols(m) ? c.st.shard.outer_global : N;
    if (Mg < 2 || Ng < 2) fail(SCANRS_ERR_SHAPE, "The input matrix must be at least 2x2.");
    if (k > std::min(Mg, Ng)) fail(SCANRS_ERR_INVALID_K, "invalid k");
    if (k == 0) fail(SCANRS_ERR_ARGUMENT, "k must be positive");
    const uint32_t l_full = (uint32_t)std::max<uint64_t>(k + 4, (uint64_t)((double)k * l_multiplier)); // rand_svd.rs:46
    // A projection wider than the matrix is short (rand_svd.rs accepts it: qr() of the wide panel returns a basis of the
    // whole range and the result is the exact truncated SVD): its leading min(m, n) columns already span that range.
    const uint32_t l = (uint32_t)std::min<uint64_t>(l_full, std::min(Mg, Ng));
    const bool rows_ge = Mg >= Ng;
    // m >= n: Omega on the cols side (n x l); Q ends on the rows side.  n > m: Omega (l x m), Q ends on the cols side.
    const uint64_t d_om = rows_ge ? N : M, d_q = rows_ge ? M : N;
    const bool om_sharded = rows_ge ? dim_sharded_cols(m) : dim_sharded_rows(m);
    const bool q_sharded = rows_ge ? dim_sharded_rows(m) : dim_sharded_cols(m);
    const bool om_to_q_transpose = !rows_ge; // A*X when m >= n
    const uint32_t ldl = even_up(l);
    double *Om = c.dev("rs_Om", (size_t)d_om * ldl);
    double *Qp = c.dev("rs_Q", (size_t)d_q * ldl);
    double *tmp = c.dev("rs_tmp", (size_t)std::max(d_om, d_q) * ldl);
    SCANRS_HIP(hipMemsetAsync(Om, 0, (size_t)d_om * ldl * 8, c.s));
    {
        if (om_sharded && !omega) fail(SCANRS_ERR_ARGUMENT, "a sharded start panel must be passed explicitly");
        if (!omega) {
            if (rows_ge)
                omega_fill_device(c.st, seed, d_om, l, Om, ldl, false);
            else
                omega_fill_device(c.st, seed, l, d_om, Om, ldl, true);
        } else if (rows_ge) { // (cols x l_full) row-major: the leading l columns
            SCANRS_HIP(hipMemcpy2DAsync(Om, (size_t)ldl * 8, omega, (size_t)l_full * 8, (size_t)l * 8, d_om, hipMemcpyHostToDevice, c.s));
            c.sync();
        } else { // (l_full x rows) row-major: the leading l rows
            double *tmp2 = c.dev("rs_omega_t", (size_t)l * d_om);
            c.h2d(tmp2, omega, (size_t)l * d_om);
            launch_transpose(c.st, tmp2, l, d_om, Om, ldl);
        }
    }
    // Q = qr(A Omega).Q   (rand_svd.rs:87 / :109)
    prepare_second_orientation(m, !om_to_q_transpose);
    mat_apply(m, om_to_q_transpose, Om, ldl, l, Qp, ldl);
    orth_cholqr(c, Qp, tmp, ldl, l, d_q, q_sharded);
    for (uint32_t it = 0; it < n_iter; it++) { // rand_svd.rs:89-92 / :111-114
        mat_apply(m, !om_to_q_transpose, Qp, ldl, l, Om, ldl);
        orth_cholqr(c, Om, tmp, ldl, l, d_om, om_sharded);
        mat_apply(m, om_to_q_transpose, Om, ldl, l, Qp, ldl);
        orth_cholqr(c, Qp, tmp, ldl, l, d_q, q_sharded);
    }
    // B = Q^T A  (l x n)  resp. B = A Q: its transpose/ itself lives on the Omega side
    mat_apply(m, !om_to_q_transpose, Qp, ldl, l, Om, ldl);
    c.sync();
    if (rows_ge) {
        ritz_finish(c, Qp, ldl, l, d_q, Om, ldl, d_om, om_sharded, k, u, s, v);
    } else {
        ritz_finish(c, Qp, ldl, l, d_q, Om, ldl, d_om, om_sharded, k, v, s, u);
        pca_dev_swap(c.st); // side S (Q) = cols: it holds V
    }
    return SCANRS_OK;
}

// ---- IRLBA (scan-rs/src/dim_red/irlba.rs:71-215), vectors on the device, small B on the host -----------------
// y <- y - X (X^T y) for the first j columns of X (rows x ldx); irlba.rs:19-22
// (the dots stay on the device: nobody on the host reads them — until round 6 they made a round trip per call)
static void orthog_dev(Ctx &c, double *y, const double *X, uint32_t ldx, uint32_t j, uint64_t rows, bool sharded) {
    if (j == 0) return;
    double *dC = c.dev("irlba_dots", std::max<uint32_t>(j, 128u)); // (one size for the whole run: j grows by one per step)
    launch_gram(c.st, X, ldx, j, y, 2, 1, rows, dC);
    if (sharded) allreduce_f64(c.st, dC, j);
    launch_gemm_nn(c.st, X, ldx, j, dC, 1, 1, rows, -1.0, 1.0, y, 2, y, 2);
}
static double norm_dev(Ctx &c, const double *y, uint64_t rows, bool sharded) {
    std::vector<double> g;
    gram_host(c, y, 2, 1, y, 2, 1, rows, sharded, g);
    return std::sqrt(g[0]);
}
static double invcheck(double x) { // irlba.rs:25-33
    const double eps2 = 2.0 * 2.220446049250313e-16;
    return x > eps2 ? 1.0 / x : 0.0;
}
// dst[:, col] (ld) <- alpha * src (ld 2)
static void set_col(Ctx &c, double *dst, uint32_t ld, uint32_t col, const double *src, uint64_t rows, double alpha) {
    launch_col_scale(c.st, dst + col, ld, src, 2, rows, alpha);
}
static void get_col(Ctx &c, const double *src, uint32_t ld, uint32_t col, double *dst, uint64_t rows) {
    launch_copy_cols(c.st, src + col, ld, dst, 2, rows, 1);
}

// tiny dense SVD of the m_b x m_b matrix B via the eigen-decomposition of B^T B and B B^T is not accurate
// enough for the residual test; use one-sided Jacobi (Hestenes) instead: B = U S V^T.
static void small_svd(std::vector<double> B, int n, std::vector<double> &U, std::vector<double> &S, std::vector<double> &Vt) {
    // work on columns of A = B (n x n), V = I
    std::vector<double> V((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1.0;
    auto col = [&](std::vector<double> &Mx, int j, int i) -> double & { return Mx[(size_t)i * n + j]; };
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < n; i++) {
                    a += col(B, p, i) * col(B, p, i);
                    b += col(B, q, i) * col(B, q, i);
                    g += col(B, p, i) * col(B, q, i);
                }
                if (g == 0.0 || std::fabs(g) <= 1e-300) continue;
                off = std::max(off, std::fabs(g) / std::sqrt(std::max(a * b, 1e-300)));
                if (std::fabs(g) <= 1e-16 * std::sqrt(a * b)) continue;
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < n; i++) {
                    const double x = col(B, p, i), y = col(B, q, i);
                    col(B, p, i) = cs * x - sn * y;
                    col(B, q, i) = sn * x + cs * y;
                    const double vx = col(V, p, i), vy = col(V, q, i);
                    col(V, p, i) = cs * vx - sn * vy;
                    col(V, q, i) = sn * vx + cs * vy;
                }
            }
        if (off < 1e-15) break;
    }
    std::vector<double> sv(n);
    std::vector<int> ord(n);
    for (int j = 0; j < n; j++) {
        double a = 0;
        for (int i = 0; i < n; i++) a += col(B, j, i) * col(B, j, i);
        sv[j] = std::sqrt(a);
        ord[j] = j;
    }
    std::sort(ord.begin(), ord.end(), [&](int x, int y) { return sv[x] > sv[y]; });
    U.assign((size_t)n * n, 0.0);
    Vt.assign((size_t)n * n, 0.0);
    S.assign(n, 0.0);
    for (int jj = 0; jj < n; jj++) {
        const int j = ord[jj];
        S[jj] = sv[j];
        for (int i = 0; i < n; i++) {
            U[(size_t)i * n + jj] = sv[j] > 0 ? col(B, j, i) / sv[j] : (i == jj ? 1.0 : 0.0);
            Vt[(size_t)jj * n + i] = col(V, j, i);
        }
    }
}

int pca_irlba(scanrs_mat *m, uint32_t nu, double tol, uint32_t maxit, const double *v0, const scanrs_snoop *snoop, double *u,
              double *s, double *v, uint32_t *mprod_out) {
    Ctx c(m);
    const uint64_t M = m->rows(), N = m->cols();
    // sharded handles: vectors on the sharded side are held by slices, their dot products and norms are all-reduced
    // (gram_host), the products that contract over that side are all-reduced inside mat_apply; everything on the other side and
    // the small bidiagonal problem are replicated and deterministic.
    const bool sh_m = dim_sharded_rows(m), sh_n = dim_sharded_cols(m);
    const uint64_t Mg = sh_m ? c.st.shard.outer_global : M, Ng = sh_n ? c.st.shard.outer_global : N;
    if (Mg < 2 || Ng < 2) fail(SCANRS_ERR_SHAPE, "The input matrix must be at least 2x2.");
    if (nu > std::min(Mg, Ng) || nu == 0) fail(SCANRS_ERR_INVALID_K, "invalid k");
    if (m->off_rank) fail(SCANRS_ERR_ARGUMENT, "irlba: LowRankOffset has no Ix1 Dot impl in the reference (low_rank_offset.rs:68-96)");
    if (maxit == 0) fail(SCANRS_ERR_ARGUMENT, "irlba: max_iter must be positive"); // the epilogue reads the last iteration's factors
    const uint32_t m_b = (uint32_t)std::min<uint64_t>(nu + 20, std::min<uint64_t>(3ull * nu, Ng)); // irlba.rs:87
    if (m_b < 4 || m_b <= nu) fail(SCANRS_ERR_INVALID_K, "invalid k");
    const uint32_t ldm = even_up(m_b);
    uint32_t mprod = 0, it = 0, j = 0, k = nu;
    double smax = -1.7976931348623157e308;
    double *V = c.dev("ir_V", (size_t)N * ldm), *W = c.dev("ir_W", (size_t)M * ldm);
    double *F = c.dev("ir_F", (size_t)N * 2), *wv = c.dev("ir_w", (size_t)M * 2), *vv = c.dev("ir_v", (size_t)N * 2);
    double *Vn = c.dev("ir_Vn", (size_t)N * ldm), *Wn = c.dev("ir_Wn", (size_t)M * ldm);
    SCANRS_HIP(hipMemsetAsync(V, 0, (size_t)N * ldm * 8, c.s));
    SCANRS_HIP(hipMemsetAsync(W, 0, (size_t)M * ldm * 8, c.s));
    SCANRS_HIP(hipMemsetAsync(F, 0, (size_t)N * 2 * 8, c.s));
    SCANRS_HIP(hipMemsetAsync(wv, 0, (size_t)M * 2 * 8, c.s));
    SCANRS_HIP(hipMemsetAsync(vv, 0, (size_t)N * 2 * 8, c.s));
    std::vector<double> B((size_t)m_b * m_b, 0.0), Us, Ss, Vts;
    {
        std::vector<double> h(N);
        if (v0)
            memcpy(h.data(), v0, N * 8); // the local slice when the columns are sharded
        else {
            // seed 0 (irlba.rs:107); a sharded handle takes its slice of the one sequential stream
            SmallRng rng(0);
            const uint64_t skip = sh_n ? c.st.shard.outer_begin : 0;
            for (uint64_t i = 0; i < skip; i++) (void)rng.normal();
            for (auto &x : h) x = rng.normal();
        }
        upload_panel(c, h.data(), N, 1, vv, 2);
        const double nn = norm_dev(c, vv, N, sh_n); // global norm
        set_col(c, V, ldm, 0, vv, N, 1.0 / nn);
    }
    double fnorm = 0.0;
    std::vector<double> resid(m_b);
    // Squares of the norms of a restart cycle, on the device: ns[j] = |w_j|^2 (B's diagonal), nf[j] = |F_j|^2 (its superdiagonal). The
    // recurrence takes them from there (col_scalar_dev_kernel) and the host reads all of them once per cycle, when it builds B — until
    // round 6 every norm was a Gram kernel, a copy and a synchronisation with the device idle in between, two per Lanczos step.
    double *ns = c.dev("ir_ns", m_b), *nf = c.dev("ir_nf", m_b);
    std::vector<double> h_ns(m_b), h_nf(m_b);
    auto norm_sq_dev = [&](const double *y, uint64_t rows, bool sharded, double *slot) {
        launch_gram(c.st, y, 2, 1, y, 2, 1, rows, slot);
        if (sharded) allreduce_f64(c.st, slot, 1);
    };
    while (it < maxit) {
        if (it > 0) j = k;
        const uint32_t j_first = j;
        get_col(c, V, ldm, j, vv, N);
        mat_apply(m, false, vv, 2, 1, wv, 2); // W[:, j] = A V[:, j]
        mprod++;
        if (it > 0) orthog_dev(c, wv, W, ldm, j, M, sh_m); // irlba.rs:131-134 (assigned to column k == j)
        norm_sq_dev(wv, M, sh_m, ns + j);
        launch_col_scale_dev(c.st, W + j, ldm, wv, 2, M, ns + j);
        while (j < m_b) {
            get_col(c, W, ldm, j, wv, M);
            mat_apply(m, true, wv, 2, 1, F, 2); // F = W[:, j]^T A
            mprod++;
            launch_col_axpy_dev(c.st, F, 2, V + j, ldm, N, ns + j); // F -= V[:, j] * s
            orthog_dev(c, F, V, ldm, j + 1, N, sh_n);
            norm_sq_dev(F, N, sh_n, nf + j);
            launch_col_scale_dev(c.st, F, 2, F, 2, N, nf + j); // F *= 1 / |F| (in place)
            if (j + 1 < m_b) {
                launch_copy_cols(c.st, F, 2, V + j + 1, ldm, N, 1);
                mat_apply(m, false, F, 2, 1, wv, 2); // A V[:, j+1] (the reference computes it twice, irlba.rs:152,155)
                mprod += 1;
                launch_col_axpy_dev(c.st, wv, 2, W + j, ldm, M, nf + j); // - W[:, j] * |F|
                orthog_dev(c, wv, W, ldm, j + 1, M, sh_m);
                norm_sq_dev(wv, M, sh_m, ns + j + 1);
                launch_col_scale_dev(c.st, W + j + 1, ldm, wv, 2, M, ns + j + 1);
            }
            j++;
        }
        // the cycle's norms, one look: B[j][j] = |w_j|, B[j][j+1] = |F_j| (irlba.rs:147-160)
        c.d2h(h_ns.data() + j_first, ns + j_first, m_b - j_first);
        c.d2h(h_nf.data() + j_first, nf + j_first, m_b - j_first);
        for (uint32_t jj = j_first; jj < m_b; jj++) {
            B[(size_t)jj * m_b + jj] = std::sqrt(h_ns[jj]);
            if (jj + 1 < m_b) B[(size_t)jj * m_b + jj + 1] = std::sqrt(h_nf[jj]);
        }
        fnorm = std::sqrt(h_nf[m_b - 1]);
        small_svd(B, (int)m_b, Us, Ss, Vts);
        for (uint32_t i = 0; i < m_b; i++) resid[i] = fnorm * Us[(size_t)(m_b - 1) * m_b + i];
        smax = Ss[0] > smax ? Ss[0] : smax;
        uint32_t num_converged = 0;
        for (uint32_t i = 0; i < nu; i++)
            if (resid[i] < tol * smax) num_converged++; // no abs(), as irlba.rs:176-180
        if (num_converged < nu) {
            k = std::max(num_converged + nu, k);
            k = std::min(k, m_b - 3);
        } else {
            break;
        }
        { // Ritz vector update, irlba.rs:190-203
            std::vector<double> Wm((size_t)m_b * k);
            for (uint32_t r = 0; r < m_b; r++)
                for (uint32_t q2 = 0; q2 < k; q2++) Wm[(size_t)r * k + q2] = Vts[(size_t)q2 * m_b + r]; // vt.t()[:, 0..k]
            double *dW = c.dev("irlba_w2", (size_t)m_b * k);
            c.h2d(dW, Wm.data(), Wm.size());
            launch_gemm_nn(c.st, V, ldm, m_b, dW, k, k, N, 1.0, 0.0, nullptr, 0, Vn, ldm);
            launch_copy_cols(c.st, Vn, ldm, V, ldm, N, k);
            set_col(c, V, ldm, k, F, N, 1.0);
            std::fill(B.begin(), B.end(), 0.0);
            for (uint32_t l2 = 0; l2 < k; l2++) B[(size_t)l2 * m_b + l2] = Ss[l2];
            for (uint32_t l2 = 0; l2 < k; l2++) B[(size_t)l2 * m_b + k] = resid[l2];
            for (uint32_t r = 0; r < m_b; r++)
                for (uint32_t q2 = 0; q2 < k; q2++) Wm[(size_t)r * k + q2] = Us[(size_t)r * m_b + q2];
            c.h2d(dW, Wm.data(), Wm.size());
            launch_gemm_nn(c.st, W, ldm, m_b, dW, k, k, M, 1.0, 0.0, nullptr, 0, Wn, ldm);
            launch_copy_cols(c.st, Wn, ldm, W, ldm, M, k);
        }
        it++;
        c.sync();
        progress_check(snoop, (double)it / (double)maxit);
    }
    {
        std::vector<double> Wm((size_t)m_b * nu);
        for (uint32_t r = 0; r < m_b; r++)
            for (uint32_t q2 = 0; q2 < nu; q2++) Wm[(size_t)r * nu + q2] = Us[(size_t)r * m_b + q2];
        double *dW = c.dev("irlba_w2", (size_t)m_b * nu);
        c.h2d(dW, Wm.data(), Wm.size());
        launch_gemm_nn(c.st, W, ldm, m_b, dW, nu, nu, M, 1.0, 0.0, nullptr, 0, Wn, ldm);
        download_panel(c, Wn, ldm, M, nu, u);
        for (uint32_t r = 0; r < m_b; r++)
            for (uint32_t q2 = 0; q2 < nu; q2++) Wm[(size_t)r * nu + q2] = Vts[(size_t)q2 * m_b + r];
        c.h2d(dW, Wm.data(), Wm.size());
        launch_gemm_nn(c.st, V, ldm, m_b, dW, nu, nu, N, 1.0, 0.0, nullptr, 0, Vn, ldm);
        download_panel(c, Vn, ldm, N, nu, v);
        for (uint32_t i = 0; i < nu; i++) s[i] = Ss[i];
    }
    if (mprod_out) *mprod_out = mprod;
    return SCANRS_OK;
}

} // namespace scanrs
