// 10x feature-barcode matrix / analysis files -> host arrays: the functions of hdf5-io/src/matrix.rs and analysis.rs
// over the library's own HDF5 parser (h5lite.cpp), and their C ABI. Host-only: no HIP in this translation unit.
#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <new>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "common_err.hpp"
#include "h5lite.hpp"
#include "host_matrix.hpp"

using scanrs::fail;
using scanrs::Failure;
namespace h5 = scanrs::h5;

namespace {

template <typename F>
int guard(F &&f) {
    try {
        f();
        return SCANRS_OK;
    } catch (const Failure &e) {
        return e.code;
    } catch (const std::bad_alloc &) {
        scanrs::set_error("out of host memory");
        return SCANRS_ERR_IO;
    } catch (const std::exception &e) {
        scanrs::set_error("internal error: %s", e.what());
        return SCANRS_ERR_IO;
    }
}

// hdf5-io/src/matrix.rs:200-205
h5::File::Object get_matrix(const h5::File &f) {
    try {
        return f.open(f.root(), "matrix");
    } catch (const Failure &) {
        const std::string why = scanrs_last_error();
        fail(SCANRS_ERR_IO, "can't find matrix in file %s %s", f.path().c_str(), why.c_str());
    }
}

// matrix.rs:207-222 (`read_features_between(name, 0, None, ..)`)
std::vector<std::string> read_features(const h5::File &f, h5::File::Object matrix, const char *name) {
    return f.read_strings(f.open(matrix, std::string("features/") + name));
}

// matrix.rs:259-275: a dataset with an empty shape means "no barcodes"
std::vector<std::string> get_barcodes(const h5::File &f, h5::File::Object matrix) {
    const h5::File::Object d = f.open(matrix, "barcodes");
    const h5::DatasetInfo di = f.info(d);
    if (di.dims.empty()) return {};
    return f.read_strings(d);
}

// matrix.rs:247-257: values are read as f64 and cast to u32 (`as`: truncating, saturating)
std::vector<uint32_t> get_values_between(const h5::File &f, h5::File::Object matrix, uint64_t start, uint64_t end) {
    const std::vector<double> v = f.read<double>(f.open(matrix, "data"), start, end);
    std::vector<uint32_t> out(v.size());
    for (size_t i = 0; i < v.size(); i++) {
        const double x = v[i];
        out[i] = !(x == x) ? 0u : x <= 0.0 ? 0u : x >= 4294967295.0 ? 4294967295u : (uint32_t)x;
    }
    return out;
}

// does the feature survive `LabelClass::remove_unlike(pattern)` (scan-types/src/label_class.rs:95-106)?
bool type_like(const std::string &type, const char *pattern) { return type.find(pattern) != std::string::npos; }

void require_features(const std::vector<std::string> &types) {
    if (types.empty()) fail(SCANRS_ERR_IO, "no features found!"); // make_labelclass_from_feature_type_vector, label_class.rs:131-134
}

// matrix.rs:56-97
void read_csc(const char *path, scanrs_h5_matrix &m) {
    const h5::File f(path);
    const h5::File::Object matrix = get_matrix(f);
    m.name = path;
    m.indptr = f.read<uint64_t>(f.open(matrix, "indptr"));
    m.indices = f.read<uint32_t>(f.open(matrix, "indices"));
    m.values = get_values_between(f, matrix, 0, UINT64_MAX);
    m.barcodes = get_barcodes(f, matrix);
    m.feature_ids = read_features(f, matrix, "id");
    m.feature_names = read_features(f, matrix, "name");
    m.feature_types = read_features(f, matrix, "feature_type");
    require_features(m.feature_types);
    if (m.feature_ids.size() != m.feature_types.size() || m.feature_names.size() != m.feature_types.size())
        fail(SCANRS_ERR_IO, "%s: feature id / name / type lengths differ", path);
    m.rows = m.feature_ids.size();
    m.cols = m.barcodes.size();
    m.nnz = m.indices.size();
    m.storage = SCANRS_CSC;
    m.has_matrix = true;
    // the structure checks of sprs' `try_new_csc`
    if (m.indptr.size() != m.cols + 1) fail(SCANRS_ERR_IO, "%s: indptr has %zu entries for %llu barcodes", path, m.indptr.size(), (unsigned long long)m.cols);
    if (m.values.size() != m.indices.size()) fail(SCANRS_ERR_IO, "%s: data and indices lengths differ", path);
    if (m.indptr.front() != 0 || m.indptr.back() != m.nnz) fail(SCANRS_ERR_IO, "%s: indptr does not span the nonzeros", path);
    bool unsorted = false;
    for (uint64_t c = 0; c < m.cols; c++) {
        if (m.indptr[c] > m.indptr[c + 1] || m.indptr[c + 1] > m.nnz) fail(SCANRS_ERR_IO, "%s: indptr is not ascending", path);
        for (uint64_t p = m.indptr[c]; p < m.indptr[c + 1]; p++) {
            if (m.indices[p] >= m.rows) fail(SCANRS_ERR_IO, "%s: feature index %u out of bounds (%llu features)", path, m.indices[p], (unsigned long long)m.rows);
            if (p > m.indptr[c] && m.indices[p] <= m.indices[p - 1]) unsorted = true;
        }
    }
    if (unsorted) { // `new_from_unsorted_csc` (matrix.rs:71-79): sort every column, then duplicates are an error
        std::vector<uint32_t> perm;
        std::vector<uint32_t> ti, tv;
        for (uint64_t c = 0; c < m.cols; c++) {
            const uint64_t s = m.indptr[c], e = m.indptr[c + 1];
            perm.resize(e - s);
            std::iota(perm.begin(), perm.end(), 0u);
            std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return m.indices[s + a] < m.indices[s + b]; });
            ti.resize(e - s);
            tv.resize(e - s);
            for (uint64_t i = 0; i < e - s; i++) {
                ti[i] = m.indices[s + perm[i]];
                tv[i] = m.values[s + perm[i]];
            }
            for (uint64_t i = 0; i < e - s; i++) {
                if (i && ti[i] == ti[i - 1]) fail(SCANRS_ERR_IO, "%s: duplicate feature index %u in barcode %llu", path, ti[i], (unsigned long long)c);
                m.indices[s + i] = ti[i];
                m.values[s + i] = tv[i];
            }
        }
    }
}

// matrix.rs:129-199 (+ compute_genes_filter :100-127)
void read_adaptive_csr(const char *path, const char *retain_like, int64_t shrink_row, scanrs_h5_matrix &m) {
    scanrs_h5_matrix csc;
    read_csc(path, csc);
    const uint64_t F = csc.rows, C = csc.cols;
    // to_csr: counting transpose, barcode indices ascending inside every feature
    std::vector<uint64_t> rp(F + 1, 0);
    for (uint32_t r : csc.indices) rp[r + 1]++;
    for (uint64_t r = 0; r < F; r++) rp[r + 1] += rp[r];
    std::vector<uint32_t> ci(csc.nnz), cv(csc.nnz);
    {
        std::vector<uint64_t> fill(rp.begin(), rp.end() - 1);
        for (uint64_t c = 0; c < C; c++)
            for (uint64_t p = csc.indptr[c]; p < csc.indptr[c + 1]; p++) {
                const uint64_t q = fill[csc.indices[p]]++;
                ci[q] = (uint32_t)c;
                cv[q] = csc.values[p];
            }
    }
    std::vector<char> drop(F, 0);
    if (retain_like)
        for (uint64_t j = 0; j < F; j++) drop[j] = !type_like(csc.feature_types[j], retain_like);
    const uint64_t min_sum = shrink_row < 0 ? 0 : (uint64_t)shrink_row;
    for (uint64_t j = 0; j < F; j++) {
        if (drop[j]) continue;
        uint64_t sum = 0; // u64 accumulation: test_compute_genes_filter_overflow (matrix.rs:336-352)
        for (uint64_t p = rp[j]; p < rp[j + 1]; p++) sum += cv[p];
        if (sum < min_sum) drop[j] = 1;
    }
    m.name = csc.name;
    m.barcodes = std::move(csc.barcodes);
    m.storage = SCANRS_CSR;
    m.cols = C;
    m.has_matrix = true;
    m.indptr.assign(1, 0);
    for (uint64_t j = 0; j < F; j++) {
        if (drop[j]) {
            m.removed.push_back(j);
            continue;
        }
        m.indices.insert(m.indices.end(), ci.begin() + rp[j], ci.begin() + rp[j + 1]);
        m.values.insert(m.values.end(), cv.begin() + rp[j], cv.begin() + rp[j + 1]);
        m.indptr.push_back(m.indices.size());
        m.feature_ids.push_back(std::move(csc.feature_ids[j]));
        m.feature_names.push_back(std::move(csc.feature_names[j]));
    }
    // `mat.feature_types` after remove_unlike: the classes that do not match the pattern are gone; features dropped for
    // their count keep their class entry in the reference's LabelClass, so only the pattern is applied here
    for (uint64_t j = 0; j < F; j++)
        if (!retain_like || type_like(csc.feature_types[j], retain_like)) m.feature_types.push_back(std::move(csc.feature_types[j]));
    m.rows = m.feature_ids.size();
    m.nnz = m.indices.size();
}

// matrix.rs:17-54
void read_metadata(const char *path, const char *retain_like, scanrs_h5_matrix &m) {
    const h5::File f(path);
    const h5::File::Object matrix = get_matrix(f);
    m.name = path;
    m.barcodes = get_barcodes(f, matrix);
    std::vector<std::string> ids = read_features(f, matrix, "id"), names = read_features(f, matrix, "name"), types = read_features(f, matrix, "feature_type");
    require_features(types);
    if (ids.size() != types.size() || names.size() != types.size()) fail(SCANRS_ERR_IO, "%s: feature id / name / type lengths differ", path);
    for (size_t j = 0; j < types.size(); j++) {
        if (retain_like && !type_like(types[j], retain_like)) {
            m.removed.push_back(j);
            continue;
        }
        m.feature_ids.push_back(std::move(ids[j]));
        m.feature_names.push_back(std::move(names[j]));
        m.feature_types.push_back(std::move(types[j]));
    }
    m.nnz = f.info(f.open(matrix, "data")).n_elements();
    m.rows = m.feature_ids.size();
    m.cols = m.barcodes.size();
}

size_t pack_strings(const std::vector<std::string> &v, char *buf, uint64_t cap) {
    size_t need = 0;
    for (const std::string &s : v) need += s.size() + 1;
    if (buf && need <= cap) {
        char *p = buf;
        for (const std::string &s : v) {
            memcpy(p, s.c_str(), s.size() + 1);
            p += s.size() + 1;
        }
    }
    return need;
}

const std::vector<std::string> *strings_of(const scanrs_h5_matrix *m, int what) {
    switch (what) {
    case 0: return &m->barcodes;
    case 1: return &m->feature_ids;
    case 2: return &m->feature_names;
    case 3: return &m->feature_types;
    default: return nullptr;
    }
}

} // namespace

extern "C" {

int scanrs_h5_read_csc_matrix(const char *path, scanrs_h5_matrix **out) {
    return guard([&] {
        if (!path || !out) fail(SCANRS_ERR_ARGUMENT, "null argument");
        std::unique_ptr<scanrs_h5_matrix> m(new scanrs_h5_matrix);
        read_csc(path, *m);
        *out = m.release();
    });
}

int scanrs_h5_read_adaptive_csr_matrix(const char *path, const char *retain_feature_like, int64_t shrink_row, scanrs_h5_matrix **out) {
    return guard([&] {
        if (!path || !out) fail(SCANRS_ERR_ARGUMENT, "null argument");
        std::unique_ptr<scanrs_h5_matrix> m(new scanrs_h5_matrix);
        read_adaptive_csr(path, retain_feature_like, shrink_row, *m);
        *out = m.release();
    });
}

int scanrs_h5_read_matrix_metadata(const char *path, const char *retain_feature_like, scanrs_h5_matrix **out) {
    return guard([&] {
        if (!path || !out) fail(SCANRS_ERR_ARGUMENT, "null argument");
        std::unique_ptr<scanrs_h5_matrix> m(new scanrs_h5_matrix);
        read_metadata(path, retain_feature_like, *m);
        *out = m.release();
    });
}

void scanrs_h5_matrix_free(scanrs_h5_matrix *m) { delete m; }

int scanrs_h5_matrix_shape(const scanrs_h5_matrix *m, uint64_t *rows, uint64_t *cols, uint64_t *nnz, int *storage) {
    return guard([&] {
        if (!m) fail(SCANRS_ERR_ARGUMENT, "null handle");
        if (rows) *rows = m->rows;
        if (cols) *cols = m->cols;
        if (nnz) *nnz = m->nnz;
        if (storage) *storage = m->storage;
    });
}

int scanrs_h5_matrix_arrays(const scanrs_h5_matrix *m, const uint64_t **indptr, const uint32_t **indices, const uint32_t **values) {
    return guard([&] {
        if (!m) fail(SCANRS_ERR_ARGUMENT, "null handle");
        if (indptr) *indptr = m->has_matrix ? m->indptr.data() : nullptr;
        if (indices) *indices = m->has_matrix ? m->indices.data() : nullptr;
        if (values) *values = m->has_matrix ? m->values.data() : nullptr;
    });
}

int scanrs_h5_matrix_n_strings(const scanrs_h5_matrix *m, int what, uint64_t *n) {
    return guard([&] {
        if (!m || !n) fail(SCANRS_ERR_ARGUMENT, "null argument");
        if (what == 4) {
            *n = 1;
            return;
        }
        const std::vector<std::string> *v = strings_of(m, what);
        if (!v) fail(SCANRS_ERR_ARGUMENT, "unknown string table %d", what);
        *n = v->size();
    });
}

const char *scanrs_h5_matrix_string(const scanrs_h5_matrix *m, int what, uint64_t i) {
    if (!m) return nullptr;
    if (what == 4) return i == 0 ? m->name.c_str() : nullptr;
    const std::vector<std::string> *v = strings_of(m, what);
    if (!v || i >= v->size()) return nullptr;
    return (*v)[i].c_str();
}

int scanrs_h5_matrix_removed(const scanrs_h5_matrix *m, const uint64_t **removed, uint64_t *n) {
    return guard([&] {
        if (!m || !n) fail(SCANRS_ERR_ARGUMENT, "null argument");
        if (removed) *removed = m->removed.data();
        *n = m->removed.size();
    });
}

// matrix.rs:270-299
int scanrs_h5_read_umi_counts(const char *path, uint32_t *out, uint64_t cap, uint64_t *n) {
    return guard([&] {
        if (!path || !n) fail(SCANRS_ERR_ARGUMENT, "null argument");
        const h5::File f(path);
        const h5::File::Object matrix = get_matrix(f);
        const std::vector<uint64_t> ptr = f.read<uint64_t>(f.open(matrix, "indptr"));
        const uint64_t n_bc = get_barcodes(f, matrix).size();
        *n = n_bc;
        if (!out || cap < n_bc) {
            if (out) fail(SCANRS_ERR_ARGUMENT, "output holds %llu counts, %llu barcodes in the file", (unsigned long long)cap, (unsigned long long)n_bc);
            return;
        }
        if (ptr.size() != n_bc + 1) fail(SCANRS_ERR_IO, "%s: indptr has %zu entries for %llu barcodes", path, ptr.size(), (unsigned long long)n_bc);
        std::fill(out, out + n_bc, 0u);
        const uint64_t stride = 2000;
        for (uint64_t index = 0; index < n_bc; index += stride) {
            const uint64_t end_index = std::min(index + stride, n_bc);
            const uint64_t b0 = ptr[index], b1 = ptr[end_index];
            if (b1 < b0) fail(SCANRS_ERR_IO, "%s: indptr is not ascending", path);
            const std::vector<uint32_t> block = get_values_between(f, matrix, b0, b1);
            if (block.size() != b1 - b0) fail(SCANRS_ERR_IO, "%s: data is shorter than indptr says", path);
            for (uint64_t k = index; k < end_index; k++) {
                if (ptr[k] < b0 || ptr[k + 1] > b1 || ptr[k] > ptr[k + 1]) fail(SCANRS_ERR_IO, "%s: indptr is not ascending", path);
                uint32_t s = 0;
                for (uint64_t p = ptr[k] - b0; p < ptr[k + 1] - b0; p++) s += block[p];
                out[k] = s;
            }
        }
    });
}

// analysis.rs:38-41
int scanrs_h5_get_clustering_keys(const char *path, char *buf, uint64_t cap, uint64_t *n_keys, uint64_t *bytes) {
    return guard([&] {
        if (!path) fail(SCANRS_ERR_ARGUMENT, "null argument");
        const h5::File f(path);
        const std::vector<std::string> names = f.member_names(f.open(f.root(), "clustering"));
        if (n_keys) *n_keys = names.size();
        const size_t need = pack_strings(names, buf, cap);
        if (bytes) *bytes = need;
    });
}

// analysis.rs:5-20
int scanrs_h5_get_clustering(const char *path, const char *key, uint16_t *num_clusters, int16_t *clusters, uint64_t cap, uint64_t *n) {
    return guard([&] {
        if (!path || !key) fail(SCANRS_ERR_ARGUMENT, "null argument");
        const h5::File f(path);
        const h5::File::Object g = f.open(f.open(f.root(), "clustering"), key);
        const std::vector<int64_t> c = f.read<int64_t>(f.open(g, "clusters"));
        const h5::File::Object nd = f.open(g, "num_clusters");
        if (!f.info(nd).dims.empty()) fail(SCANRS_ERR_IO, "%s: num_clusters of '%s' is not a scalar", path, key);
        const std::vector<int64_t> nc = f.read<int64_t>(nd);
        if (nc.size() != 1) fail(SCANRS_ERR_IO, "%s: num_clusters of '%s' is not a scalar", path, key);
        if (num_clusters) *num_clusters = (uint16_t)nc[0];
        if (n) *n = c.size();
        if (clusters) {
            if (cap < c.size()) fail(SCANRS_ERR_ARGUMENT, "output holds %llu labels, %zu in the file", (unsigned long long)cap, c.size());
            for (size_t i = 0; i < c.size(); i++) clusters[i] = (int16_t)c[i];
        }
    });
}

// analysis.rs:23-36
int scanrs_h5_get_differential_expression(const char *path, const char *key, double *out, uint64_t cap, uint64_t *rows, uint64_t *cols) {
    return guard([&] {
        if (!path || !key) fail(SCANRS_ERR_ARGUMENT, "null argument");
        const h5::File f(path);
        const h5::File::Object d = f.open(f.open(f.open(f.root(), "all_differential_expression"), key), "data");
        const h5::DatasetInfo di = f.info(d);
        if (di.dims.size() != 2) fail(SCANRS_ERR_IO, "%s: differential expression table of '%s' is not two-dimensional", path, key);
        if (rows) *rows = di.dims[0];
        if (cols) *cols = di.dims[1];
        if (out) {
            if (cap < di.dims[0] * di.dims[1]) fail(SCANRS_ERR_ARGUMENT, "output too small for the %llu x %llu table", (unsigned long long)di.dims[0], (unsigned long long)di.dims[1]);
            const std::vector<double> v = f.read<double>(d);
            if (!v.empty()) memcpy(out, v.data(), v.size() * 8);
        }
    });
}

int scanrs_h5_read_f64(const char *path, const char *dataset, double *out, uint64_t cap, uint64_t *dims, uint32_t *rank) {
    return guard([&] {
        if (!path || !dataset) fail(SCANRS_ERR_ARGUMENT, "null argument");
        const h5::File f(path);
        const h5::File::Object d = f.open(f.root(), dataset);
        const h5::DatasetInfo di = f.info(d);
        if (di.dims.size() > 8) fail(SCANRS_ERR_IO, "%s: rank %zu", path, di.dims.size());
        if (rank) *rank = (uint32_t)di.dims.size();
        if (dims)
            for (size_t i = 0; i < di.dims.size(); i++) dims[i] = di.dims[i];
        if (out) {
            const std::vector<double> v = f.read<double>(d);
            if (cap < v.size()) fail(SCANRS_ERR_ARGUMENT, "output holds %llu values, dataset has %zu", (unsigned long long)cap, v.size());
            if (!v.empty()) memcpy(out, v.data(), v.size() * 8);
        }
    });
}

int scanrs_h5_read_strings(const char *path, const char *dataset, char *buf, uint64_t cap, uint64_t *n, uint64_t *bytes) {
    return guard([&] {
        if (!path || !dataset) fail(SCANRS_ERR_ARGUMENT, "null argument");
        const h5::File f(path);
        const std::vector<std::string> v = f.read_strings(f.open(f.root(), dataset));
        if (n) *n = v.size();
        const size_t need = pack_strings(v, buf, cap);
        if (bytes) *bytes = need;
    });
}

int scanrs_h5_member_names(const char *path, const char *group, char *buf, uint64_t cap, uint64_t *n, uint64_t *bytes) {
    return guard([&] {
        if (!path || !group) fail(SCANRS_ERR_ARGUMENT, "null argument");
        const h5::File f(path);
        const std::vector<std::string> v = f.member_names(f.open(f.root(), group));
        if (n) *n = v.size();
        const size_t need = pack_strings(v, buf, cap);
        if (bytes) *bytes = need;
    });
}

} // extern "C"
