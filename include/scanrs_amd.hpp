// scanrs_amd.hpp — header-only C++ mirror of the reference's operator / solver surface over the C ABI
// (scanrs_amd.h). Same names, argument meaning and error behaviour as the Rust API it stands for:
//   sqz::AdaptiveMat / LowRankOffset      -> scanrs::AdaptiveMat   (sqz/src/mat.rs:34-42, low_rank_offset.rs:12-16)
//   scan_rs::normalization::*              -> scanrs::normalize ... (scan-rs/src/normalization.rs:11-213)
//   scan_rs::dim_red::{BkSvd,RandSvd,Irlba} -> scanrs::BkSvd ...   (scan-rs/src/dim_red/*.rs)
//   snoop::CancelProgress                  -> scanrs::Snoop         (snoop/src/lib.rs:20-58)
// `anyhow::Error` becomes scanrs::Error (code + the reference's message), `CancellationError` its subclass.
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "scanrs_amd.h"

namespace scanrs {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
struct CancellationError : Error { // snoop/src/lib.rs:5-18
    CancellationError() : Error(SCANRS_ERR_CANCELLED, "cancellation error") {}
};
inline void check(int code) {
    if (code == SCANRS_OK) return;
    if (code == SCANRS_ERR_CANCELLED) throw CancellationError();
    throw Error(code, scanrs_last_error());
}

enum class Storage : int { CSR = SCANRS_CSR, CSC = SCANRS_CSC };                       // sqz/src/mat.rs:45-65
enum class Normalization : int {                                                       // normalization.rs:11-28
    CellRanger = 0, CellRanger8, SeuratLog, BinomialDeviance, BinomialPearson, WithSizeFactors, LogTransform
};
enum class LogBase : int { E = SCANRS_FN_LN_1P, Two = SCANRS_FN_LOG2_1P, Ten = SCANRS_FN_LOG10_1P }; // :105-112

inline Normalization normalization_from_str(const std::string &s) { // impl FromStr, normalization.rs:30-43
    if (s == "cellranger") return Normalization::CellRanger;
    if (s == "cellranger8") return Normalization::CellRanger8;
    if (s == "seuratlog") return Normalization::SeuratLog;
    if (s == "binomialdeviance") return Normalization::BinomialDeviance;
    if (s == "binomialpearson") return Normalization::BinomialPearson;
    throw Error(SCANRS_ERR_ARGUMENT, "Normalization not recognized: " + s);
}

// row-major dense array, the stand-in for ndarray::Array2<f64>
struct Array2 {
    size_t rows = 0, cols = 0;
    std::vector<double> data;
    Array2() = default;
    Array2(size_t r, size_t c) : rows(r), cols(c), data(r * c, 0.0) {}
    double &operator()(size_t r, size_t c) { return data[r * cols + c]; }
    double operator()(size_t r, size_t c) const { return data[r * cols + c]; }
};

// snoop::CancelProgress: a cancel flag another thread may set + a progress sink
struct Snoop {
    std::atomic<uint8_t> cancelled{0};
    std::function<void(double)> on_progress;
    void cancel() { cancelled.store(1, std::memory_order_relaxed); }
    bool is_cancelled() const { return cancelled.load(std::memory_order_relaxed) != 0; }
};

struct PcaResult { // (u, d, v): dim_red/mod.rs:47
    Array2 u;
    std::vector<double> s;
    Array2 v;
};

// Device-resident AdaptiveMat; plays LowRankOffset once an offset is installed.
class AdaptiveMat {
    scanrs_mat *h_ = nullptr;
    explicit AdaptiveMat(scanrs_mat *h) : h_(h) {}

  public:
    AdaptiveMat() = default;
    AdaptiveMat(const AdaptiveMat &) = delete;
    AdaptiveMat &operator=(const AdaptiveMat &) = delete;
    AdaptiveMat(AdaptiveMat &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    AdaptiveMat &operator=(AdaptiveMat &&o) noexcept {
        if (this != &o) {
            scanrs_mat_free(h_);
            h_ = o.h_;
            o.h_ = nullptr;
        }
        return *this;
    }
    ~AdaptiveMat() { scanrs_mat_free(h_); } // Drop
    scanrs_mat *raw() const { return h_; }

    // AdaptiveMat::from_csmat (mat.rs:92-124)
    static AdaptiveMat from_csmat(uint64_t rows, uint64_t cols, Storage storage, const uint64_t *indptr, const uint32_t *indices,
                                  const uint32_t *data) {
        scanrs_mat *h = nullptr;
        check(scanrs_mat_create(rows, cols, (int)storage, indptr, indices, data, &h));
        return AdaptiveMat(h);
    }
    // AdaptiveMat::new(rows, cols, storage, Vec<AdaptiveVec>) (mat.rs:68-90): the encoded vectors, decoded on the device
    static AdaptiveMat from_adaptive_vecs(uint64_t rows, uint64_t cols, Storage storage, const std::vector<scanrs_adaptive_vec> &vecs) {
        scanrs_mat *h = nullptr;
        check(scanrs_mat_create_adaptive(rows, cols, (int)storage, vecs.data(), vecs.size(), &h));
        return AdaptiveMat(h);
    }
    AdaptiveMat view() const { // mat.rs:242-245
        scanrs_mat *h = nullptr;
        check(scanrs_mat_view(h_, &h));
        return AdaptiveMat(h);
    }
    AdaptiveMat t() const { // mat.rs:262-270 / low_rank_offset.rs:60-65
        scanrs_mat *h = nullptr;
        check(scanrs_mat_t(h_, &h));
        return AdaptiveMat(h);
    }
    uint64_t rows() const { return shape().first; }
    uint64_t cols() const { return shape().second; }
    std::pair<uint64_t, uint64_t> shape() const {
        uint64_t r = 0, c = 0;
        check(scanrs_mat_shape(h_, &r, &c));
        return {r, c};
    }
    uint64_t nnz() const {
        uint64_t n = 0;
        check(scanrs_mat_nnz(h_, &n));
        return n;
    }
    // lazy maps (consume-and-return in the reference; in place here)
    AdaptiveMat &compose_scale_axis(int axis, const std::vector<double> &factors) {
        if (factors.size() != (axis == 0 ? rows() : cols())) throw Error(SCANRS_ERR_SHAPE, "Dimension mismatch");
        check(scanrs_mat_compose_scale_axis(h_, axis, factors.data()));
        return *this;
    }
    AdaptiveMat &apply(int scalar_fn) {
        check(scanrs_mat_apply(h_, scalar_fn));
        return *this;
    }
    AdaptiveMat &scale_and_center(int axis, const std::vector<double> *scaling = nullptr) { // mat.rs:986-1001
        check(scanrs_mat_scale_and_center(h_, axis, scaling ? scaling->data() : nullptr));
        return *this;
    }
    AdaptiveMat &center(int axis, const std::vector<double> *means = nullptr) {
        check(scanrs_mat_center(h_, axis, means ? means->data() : nullptr));
        return *this;
    }
    AdaptiveMat &scale(int axis, const std::vector<double> *stds = nullptr) {
        check(scanrs_mat_scale(h_, axis, stds ? stds->data() : nullptr));
        return *this;
    }
    // reductions
    std::vector<uint32_t> sum_axis_u32(int axis) const {
        std::vector<uint32_t> out(axis == 0 ? cols() : rows());
        check(scanrs_mat_sum_axis_u32(h_, axis, out.data()));
        return out;
    }
    std::vector<double> mean_axis(int axis) const {
        std::vector<double> out(axis == 0 ? cols() : rows());
        check(scanrs_mat_mean_axis(h_, axis, out.data()));
        return out;
    }
    // Dot impls: self.dot(rhs) and lhs.dot(self)
    Array2 dot(const Array2 &rhs) const {
        if (rhs.rows != cols()) throw Error(SCANRS_ERR_SHAPE, "Dimension mismatch"); // prod.rs:41
        Array2 out(rows(), rhs.cols);
        check(scanrs_mat_dot(h_, rhs.data.data(), (uint32_t)rhs.cols, out.data.data()));
        return out;
    }
    Array2 rdot(const Array2 &lhs) const {
        if (lhs.cols != rows()) throw Error(SCANRS_ERR_SHAPE, "Dimension mismatch");
        Array2 out(lhs.rows, cols());
        check(scanrs_mat_rdot(h_, lhs.data.data(), (uint32_t)lhs.rows, out.data.data()));
        return out;
    }
    Array2 to_dense() const {
        Array2 out(rows(), cols());
        check(scanrs_mat_to_dense(h_, out.data.data()));
        return out;
    }
};

// normalize(mat, norm) -> LowRankOffset (normalization.rs:46-69); consumes `mat` like the reference.
inline AdaptiveMat normalize(AdaptiveMat mat, Normalization norm) {
    check(scanrs_normalize(mat.raw(), (int)norm, nullptr));
    return mat;
}
inline AdaptiveMat normalize_with_size_factor(AdaptiveMat mat, Normalization norm, const std::vector<uint32_t> *size_factors) {
    if (size_factors && size_factors->size() != mat.cols())
        throw Error(SCANRS_ERR_SHAPE, "Size of the size factor and matrix columns dont match.");
    check(scanrs_normalize(mat.raw(), (int)norm, size_factors ? size_factors->data() : nullptr));
    return mat;
}
inline AdaptiveMat binom_deviance_resid(AdaptiveMat mat) { return normalize(std::move(mat), Normalization::BinomialDeviance); }
inline AdaptiveMat binom_pearson_resid(AdaptiveMat mat) { return normalize(std::move(mat), Normalization::BinomialPearson); }

// nn::knn (scan-rs/src/nn.rs:38-57): k nearest other rows of the cells x d matrix, nearest first
inline std::vector<uint32_t> knn(const Array2 &v, size_t k) {
    std::vector<uint32_t> out(v.rows * k);
    check(scanrs_knn(v.data.data(), v.rows, (uint32_t)v.cols, (uint32_t)k, out.data()));
    return out;
}

namespace detail {
inline void progress_tramp(void *ctx, double f) {
    auto *s = static_cast<Snoop *>(ctx);
    if (s->on_progress) s->on_progress(f);
}
inline scanrs_snoop make_snoop(Snoop *s) {
    scanrs_snoop sn;
    sn.cancel = reinterpret_cast<const volatile uint8_t *>(&s->cancelled);
    sn.progress = &progress_tramp;
    sn.ctx = s;
    return sn;
}
} // namespace detail

// trait Pca<T, N> (dim_red/mod.rs:103-111): run_pca_cancellable / run_pca
struct BkSvd { // dim_red/bk_svd.rs:16-39
    double k_multiplier = 2.0;
    size_t n_iter = 5;
    PcaResult run_pca_cancellable(const AdaptiveMat &m, size_t k, Snoop *snoop) const {
        PcaResult r{Array2(m.rows(), k), std::vector<double>(k), Array2(m.cols(), k)};
        scanrs_snoop sn;
        if (snoop) sn = detail::make_snoop(snoop);
        check(scanrs_pca_bk(m.raw(), (uint32_t)k, k_multiplier, (uint32_t)n_iter, 0, nullptr, snoop ? &sn : nullptr,
                            r.u.data.data(), r.s.data(), r.v.data.data()));
        return r;
    }
    PcaResult run_pca(const AdaptiveMat &m, size_t k) const { return run_pca_cancellable(m, k, nullptr); }
};
struct RandSvd { // dim_red/rand_svd.rs:13-35
    double l_multiplier = 10.0;
    size_t n_iter = 2;
    PcaResult run_pca(const AdaptiveMat &m, size_t k) const {
        PcaResult r{Array2(m.rows(), k), std::vector<double>(k), Array2(m.cols(), k)};
        check(scanrs_pca_rand(m.raw(), (uint32_t)k, l_multiplier, (uint32_t)n_iter, 0, nullptr, r.u.data.data(), r.s.data(),
                              r.v.data.data()));
        return r;
    }
};
struct Irlba { // dim_red/irlba.rs:36-57
    double tol = 0.0001;
    size_t max_iter = 50;
    PcaResult run_pca_cancellable(const AdaptiveMat &m, size_t k, Snoop *snoop) const {
        PcaResult r{Array2(m.rows(), k), std::vector<double>(k), Array2(m.cols(), k)};
        scanrs_snoop sn;
        if (snoop) sn = detail::make_snoop(snoop);
        uint32_t mprod = 0;
        check(scanrs_pca_irlba(m.raw(), (uint32_t)k, tol, (uint32_t)max_iter, nullptr, snoop ? &sn : nullptr, r.u.data.data(),
                               r.s.data(), r.v.data.data(), &mprod));
        return r;
    }
    PcaResult run_pca(const AdaptiveMat &m, size_t k) const { return run_pca_cancellable(m, k, nullptr); }
};

// PcaResult left in device memory by the last BkSvd / RandSvd call on `m` (scanrs_pca_result_device): for device consumers
struct PcaResultDevice {
    const double *d_u = nullptr, *d_v = nullptr; // rows x k, cols x k, row-major with leading dimensions ld_u / ld_v (elements)
    uint32_t ld_u = 0, ld_v = 0, k = 0;
};
inline PcaResultDevice pca_result_device(const AdaptiveMat &m) {
    PcaResultDevice r;
    check(scanrs_pca_result_device(m.raw(), &r.d_u, &r.ld_u, &r.d_v, &r.ld_v, &r.k));
    return r;
}
// nn::knn on scores that are already in device memory
inline std::vector<uint32_t> knn_device(const double *d_points, size_t n, uint32_t ld, uint32_t d, size_t k) {
    std::vector<uint32_t> out(n * k);
    check(scanrs_knn_device(d_points, n, ld, d, (uint32_t)k, out.data()));
    return out;
}

// The single-process multi-GPU form (scanrs_multi_*): one object for the whole matrix, sharded by the library over
// `n_shards` devices; normalize + run_pca as on a single handle, results for the whole matrix.
class MultiMat {
    scanrs_multi *h_ = nullptr;
    size_t rows_ = 0, cols_ = 0;

  public:
    MultiMat(size_t rows, size_t cols, int storage, const uint64_t *indptr, const uint32_t *indices, const uint32_t *values,
             uint32_t n_shards, const int *devices = nullptr)
        : rows_(rows), cols_(cols) {
        check(scanrs_multi_create(rows, cols, storage, indptr, indices, values, n_shards, devices, &h_));
    }
    MultiMat(const MultiMat &) = delete;
    MultiMat &operator=(const MultiMat &) = delete;
    ~MultiMat() { scanrs_multi_free(h_); }
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    void normalize(Normalization norm) { check(scanrs_multi_normalize(h_, (int)norm, nullptr)); }
    PcaResult run_pca(const BkSvd &cfg, size_t k) {
        PcaResult r{Array2(rows_, k), std::vector<double>(k), Array2(cols_, k)};
        check(scanrs_multi_pca_bk(h_, (uint32_t)k, cfg.k_multiplier, (uint32_t)cfg.n_iter, 0, nullptr, nullptr, r.u.data.data(),
                                  r.s.data(), r.v.data.data()));
        return r;
    }
};

// ---- hdf5-io crate (hdf5-io/src/matrix.rs, analysis.rs): 10x files -> host arrays, parsed by the library itself ----
namespace hdf5_io {
static const char *const FEATURE_TYPE_GENE_EXPRESSION = "Gene Expression"; // matrix.rs:14

// GenericFeatureBarcodeMatrix / MatrixMetadata (scan-types/src/matrix.rs:8-15) with the matrix as host arrays
struct FeatureBarcodeMatrix {
    std::string name;
    std::vector<std::string> barcodes, feature_ids, feature_names, feature_types;
    Storage storage = Storage::CSC;
    uint64_t rows = 0, cols = 0, nnz = 0;
    std::vector<uint64_t> indptr; // empty for read_matrix_metadata
    std::vector<uint32_t> indices, values;
    std::vector<uint64_t> removed_features; // ascending (the BTreeSet of the reference)
    AdaptiveMat to_device() const { return AdaptiveMat::from_csmat(rows, cols, storage, indptr.data(), indices.data(), values.data()); }
};

namespace detail {
inline FeatureBarcodeMatrix take(scanrs_h5_matrix *h) {
    struct Free {
        scanrs_h5_matrix *h;
        ~Free() { scanrs_h5_matrix_free(h); }
    } guard{h};
    FeatureBarcodeMatrix m;
    int storage = 0;
    check(scanrs_h5_matrix_shape(h, &m.rows, &m.cols, &m.nnz, &storage));
    m.storage = (Storage)storage;
    auto strings = [&](int what) {
        uint64_t n = 0;
        check(scanrs_h5_matrix_n_strings(h, what, &n));
        std::vector<std::string> v(n);
        for (uint64_t i = 0; i < n; i++) v[i] = scanrs_h5_matrix_string(h, what, i);
        return v;
    };
    m.barcodes = strings(0);
    m.feature_ids = strings(1);
    m.feature_names = strings(2);
    m.feature_types = strings(3);
    m.name = strings(4)[0];
    const uint64_t *ip = nullptr, *rem = nullptr;
    const uint32_t *ix = nullptr, *vv = nullptr;
    check(scanrs_h5_matrix_arrays(h, &ip, &ix, &vv));
    if (ip) {
        const uint64_t n_outer = m.storage == Storage::CSR ? m.rows : m.cols;
        m.indptr.assign(ip, ip + n_outer + 1);
        m.indices.assign(ix, ix + m.nnz);
        m.values.assign(vv, vv + m.nnz);
    }
    uint64_t n_rem = 0;
    check(scanrs_h5_matrix_removed(h, &rem, &n_rem));
    m.removed_features.assign(rem, rem + n_rem);
    return m;
}
inline std::vector<std::string> unpack(const std::vector<char> &buf, uint64_t n) {
    std::vector<std::string> out;
    const char *p = buf.data();
    for (uint64_t i = 0; i < n; i++) {
        out.emplace_back(p);
        p += out.back().size() + 1;
    }
    return out;
}
} // namespace detail

inline FeatureBarcodeMatrix read_csc_matrix(const std::string &path) { // matrix.rs:56-97
    scanrs_h5_matrix *h = nullptr;
    check(scanrs_h5_read_csc_matrix(path.c_str(), &h));
    return detail::take(h);
}
// matrix.rs:129-199; retain_feature_like == nullptr: None; shrink_row < 0: None
inline FeatureBarcodeMatrix read_adaptive_csr_matrix(const std::string &path, const char *retain_feature_like = nullptr, int64_t shrink_row = -1) {
    scanrs_h5_matrix *h = nullptr;
    check(scanrs_h5_read_adaptive_csr_matrix(path.c_str(), retain_feature_like, shrink_row, &h));
    return detail::take(h);
}
inline FeatureBarcodeMatrix read_matrix_metadata(const std::string &path, const char *retain_feature_like = nullptr) { // matrix.rs:17-54
    scanrs_h5_matrix *h = nullptr;
    check(scanrs_h5_read_matrix_metadata(path.c_str(), retain_feature_like, &h));
    return detail::take(h);
}
inline std::vector<uint32_t> read_umi_counts_from_matrix(const std::string &path) { // matrix.rs:270-299
    uint64_t n = 0;
    check(scanrs_h5_read_umi_counts(path.c_str(), nullptr, 0, &n));
    std::vector<uint32_t> out(n);
    check(scanrs_h5_read_umi_counts(path.c_str(), out.data(), n, &n));
    return out;
}
inline std::vector<std::string> get_clustering_keys(const std::string &analysis_h5) { // analysis.rs:38-41
    uint64_t n = 0, bytes = 0;
    check(scanrs_h5_get_clustering_keys(analysis_h5.c_str(), nullptr, 0, &n, &bytes));
    std::vector<char> buf(bytes + 1);
    check(scanrs_h5_get_clustering_keys(analysis_h5.c_str(), buf.data(), bytes, &n, &bytes));
    return detail::unpack(buf, n);
}
inline std::pair<uint16_t, std::vector<int16_t>> get_clustering(const std::string &analysis_h5, const std::string &key) { // analysis.rs:5-20
    uint16_t nc = 0;
    uint64_t n = 0;
    check(scanrs_h5_get_clustering(analysis_h5.c_str(), key.c_str(), &nc, nullptr, 0, &n));
    std::vector<int16_t> c(n);
    check(scanrs_h5_get_clustering(analysis_h5.c_str(), key.c_str(), &nc, c.data(), n, &n));
    return {nc, std::move(c)};
}
inline Array2 get_differential_expression(const std::string &analysis_h5, const std::string &key) { // analysis.rs:23-36
    uint64_t r = 0, c = 0;
    check(scanrs_h5_get_differential_expression(analysis_h5.c_str(), key.c_str(), nullptr, 0, &r, &c));
    Array2 out(r, c);
    check(scanrs_h5_get_differential_expression(analysis_h5.c_str(), key.c_str(), out.data.data(), r * c, &r, &c));
    return out;
}
} // namespace hdf5_io

namespace mtx {
// scan_rs::mtx::load_mtx (scan-rs/src/mtx.rs:10-51): the CSR arrays; `.to_device()` is the AdaptiveMat the reference returns
inline hdf5_io::FeatureBarcodeMatrix read_mtx(const std::string &path) {
    scanrs_h5_matrix *h = nullptr;
    check(scanrs_mtx_read(path.c_str(), &h));
    return hdf5_io::detail::take(h);
}
inline AdaptiveMat load_mtx(const std::string &path) { return read_mtx(path).to_device(); }
} // namespace mtx

} // namespace scanrs
