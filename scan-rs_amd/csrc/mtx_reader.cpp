// `load_mtx` of scan-rs/src/mtx.rs:10-51: gzipped MatrixMarket coordinate file -> CSR arrays (what AdaptiveMat::from_csmat
// takes). Host-only. Kept line for line with the reference's parser: '%' lines are comments, the first other line is
// "NROW NCOL NNZ", every further line is "ROW COL VAL" (1-based, VAL a u32), duplicates are summed (TriMat::to_csr),
// column indices ascend inside a row. Where the reference panics (index 0 or past the shape) this returns an error.
// zlib's gz* layer reads concatenated gzip members like flate2's MultiGzDecoder; it also reads a plain-text file, which
// the reference rejects — the only deliberate difference.
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>

#include "common_err.hpp"
#include "host_matrix.hpp"

using scanrs::fail;
using scanrs::Failure;

namespace {

// str::parse::<usize/u32>: optional '+', decimal digits only, no overflow
bool parse_uint(const char *b, const char *e, uint64_t max, uint64_t &out) {
    if (b < e && *b == '+') b++;
    if (b == e) return false;
    uint64_t v = 0;
    for (; b < e; b++) {
        if (*b < '0' || *b > '9') return false;
        const uint64_t d = (uint64_t)(*b - '0');
        if (v > (max - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

// next whitespace-separated token of [p, end)
bool next_token(const char *&p, const char *end, const char *&tb, const char *&te) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\v' || *p == '\f')) p++;
    if (p >= end) return false;
    tb = p;
    while (p < end && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\v' || *p == '\f')) p++;
    te = p;
    return true;
}

void read_mtx(const char *path, scanrs_h5_matrix &m) {
    gzFile f = gzopen(path, "rb");
    if (!f) fail(SCANRS_ERR_IO, "%s: %s", path, strerror(errno)); // `.with_context(|| path.display().to_string())`
    struct Close {
        gzFile f;
        ~Close() { gzclose(f); }
    } closer{f};
    gzbuffer(f, 1u << 20);
    std::string line;
    std::vector<char> buf(1 << 16);
    bool have_header = false;
    uint64_t nrow = 0, ncol = 0;
    std::vector<uint32_t> tr, tc, tv;
    for (;;) {
        line.clear();
        bool got = false;
        while (gzgets(f, buf.data(), (int)buf.size())) { // a line may be longer than the buffer
            got = true;
            line.append(buf.data());
            if (!line.empty() && line.back() == '\n') break;
        }
        if (!got) {
            int errnum = 0;
            const char *msg = gzerror(f, &errnum);
            if (errnum != Z_OK && errnum != Z_STREAM_END && !gzeof(f)) fail(SCANRS_ERR_IO, "%s: %s", path, msg ? msg : "read error");
            break;
        }
        if (line[0] == '%') continue;
        const char *p = line.data(), *end = line.data() + line.size(), *tb = nullptr, *te = nullptr;
        uint64_t a = 0, b = 0, c = 0;
        if (!have_header) {
            if (!next_token(p, end, tb, te)) fail(SCANRS_ERR_IO, "no NROW");
            if (!parse_uint(tb, te, UINT64_MAX, a)) fail(SCANRS_ERR_IO, "invalid digit found in string");
            if (!next_token(p, end, tb, te)) fail(SCANRS_ERR_IO, "no NCOL");
            if (!parse_uint(tb, te, UINT64_MAX, b)) fail(SCANRS_ERR_IO, "invalid digit found in string");
            if (!next_token(p, end, tb, te)) fail(SCANRS_ERR_IO, "no NNZ");
            if (!parse_uint(tb, te, UINT64_MAX, c)) fail(SCANRS_ERR_IO, "invalid digit found in string");
            if (a > 0xFFFFFFFFull || b > 0xFFFFFFFFull) fail(SCANRS_ERR_SHAPE, "dimensions must fit in u32 (AdaptiveVec limit)");
            nrow = a;
            ncol = b;
            const size_t cap = (size_t)std::min<uint64_t>(c, 1ull << 27); // with_capacity: a hint, not a promise (a corrupt header must not reserve gigabytes)
            tr.reserve(cap);
            tc.reserve(cap);
            tv.reserve(cap);
            have_header = true;
            continue;
        }
        if (!next_token(p, end, tb, te)) fail(SCANRS_ERR_IO, "missing ROW");
        if (!parse_uint(tb, te, UINT64_MAX, a)) fail(SCANRS_ERR_IO, "invalid digit found in string");
        if (!next_token(p, end, tb, te)) fail(SCANRS_ERR_IO, "missing COL");
        if (!parse_uint(tb, te, UINT64_MAX, b)) fail(SCANRS_ERR_IO, "invalid digit found in string");
        if (!next_token(p, end, tb, te)) fail(SCANRS_ERR_IO, "missing VAL");
        if (!parse_uint(tb, te, 0xFFFFFFFFull, c)) fail(SCANRS_ERR_IO, "invalid digit found in string"); // parse::<u32>
        if (a < 1 || a > nrow || b < 1 || b > ncol) fail(SCANRS_ERR_IO, "%s: triplet (%llu, %llu) outside the %llu x %llu matrix", path,
                                                            (unsigned long long)a, (unsigned long long)b, (unsigned long long)nrow, (unsigned long long)ncol);
        tr.push_back((uint32_t)(a - 1));
        tc.push_back((uint32_t)(b - 1));
        tv.push_back((uint32_t)c);
    }
    if (!have_header) fail(SCANRS_ERR_IO, "no matrix found");
    // TriMat::to_csr: counting sort by row, columns ascending inside a row, duplicates summed (u32, wrapping like a release build)
    std::vector<uint64_t> start(nrow + 1, 0);
    for (uint32_t r : tr) start[r + 1]++;
    for (uint64_t i = 0; i < nrow; i++) start[i + 1] += start[i];
    std::vector<uint64_t> fill(start.begin(), start.end() - 1);
    std::vector<std::pair<uint32_t, uint32_t>> ent(tr.size());
    for (size_t i = 0; i < tr.size(); i++) ent[fill[tr[i]]++] = {tc[i], tv[i]};
    m.indptr.assign(nrow + 1, 0);
    m.indices.reserve(ent.size());
    m.values.reserve(ent.size());
    for (uint64_t r = 0; r < nrow; r++) {
        std::stable_sort(ent.begin() + start[r], ent.begin() + start[r + 1],
                         [](const std::pair<uint32_t, uint32_t> &x, const std::pair<uint32_t, uint32_t> &y) { return x.first < y.first; });
        for (uint64_t q = start[r]; q < start[r + 1]; q++) {
            if (m.indices.size() > m.indptr[r] && m.indices.back() == ent[q].first)
                m.values.back() += ent[q].second;
            else {
                m.indices.push_back(ent[q].first);
                m.values.push_back(ent[q].second);
            }
        }
        m.indptr[r + 1] = m.indices.size();
    }
    m.name = path;
    m.rows = nrow;
    m.cols = ncol;
    m.nnz = m.indices.size();
    m.storage = SCANRS_CSR;
    m.has_matrix = true;
}

} // namespace

extern "C" int scanrs_mtx_read(const char *path, scanrs_h5_matrix **out) {
    try {
        if (!path || !out) fail(SCANRS_ERR_ARGUMENT, "null argument");
        std::unique_ptr<scanrs_h5_matrix> m(new scanrs_h5_matrix);
        read_mtx(path, *m);
        *out = m.release();
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
