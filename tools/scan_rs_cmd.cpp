// scan-rs-cmd on MI355X: the reference's command line tool (tools/src/bin/cmd.rs:16-104) over the
// C++ mirror of its API (include/scanrs_amd.hpp):
//   scan-rs-cmd INPUT.mtx[.gz] -o OUT_DIR -n {cellranger|cellranger8|seuratlog|binomialdeviance|binomialpearson} -d NUM_PCS
// (INPUT may also be a 10x feature-barcode matrix .h5 — read through the hdf5-io mirror; the reference CLI takes mtx only)
// writes svd_u.csv.gz, svd_d.csv.gz, svd_v.csv.gz exactly as cmd.rs:83-86 does.
#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <vector>
#include <zlib.h>

#include "scanrs_amd.hpp"

using namespace scanrs;

// load_mtx (scan-rs/src/mtx.rs:10-51): gz or plain MatrixMarket coordinate file -> CSR AdaptiveMat
// (TriMat::to_csr: duplicates are summed, indices ascending).
static AdaptiveMat load_mtx(const std::string &path) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw Error(SCANRS_ERR_ARGUMENT, path);
    std::vector<char> line(1 << 16);
    bool have_header = false;
    size_t nrow = 0, ncol = 0, nnz = 0;
    std::vector<uint32_t> tr, tc, tv;
    while (gzgets(f, line.data(), (int)line.size())) {
        if (line[0] == '%') continue;
        char *p = line.data();
        char *e = nullptr;
        if (!have_header) {
            nrow = strtoull(p, &e, 10);
            if (e == p) continue; // blank line
            p = e;
            ncol = strtoull(p, &e, 10);
            if (e == p) throw Error(SCANRS_ERR_ARGUMENT, "no NCOL");
            p = e;
            nnz = strtoull(p, &e, 10);
            if (e == p) throw Error(SCANRS_ERR_ARGUMENT, "no NNZ");
            tr.reserve(nnz);
            tc.reserve(nnz);
            tv.reserve(nnz);
            have_header = true;
            continue;
        }
        const unsigned long long r = strtoull(p, &e, 10);
        if (e == p) continue;
        p = e;
        const unsigned long long c = strtoull(p, &e, 10);
        if (e == p) throw Error(SCANRS_ERR_ARGUMENT, "missing COL");
        p = e;
        const unsigned long long v = strtoull(p, &e, 10);
        if (e == p) throw Error(SCANRS_ERR_ARGUMENT, "missing VAL");
        if (r < 1 || r > nrow || c < 1 || c > ncol) throw Error(SCANRS_ERR_ARGUMENT, "triplet out of range");
        tr.push_back((uint32_t)(r - 1));
        tc.push_back((uint32_t)(c - 1));
        tv.push_back((uint32_t)v);
    }
    gzclose(f);
    if (!have_header) throw Error(SCANRS_ERR_ARGUMENT, "no matrix found");
    // counting sort by row, then sort columns inside each row and merge duplicates
    std::vector<uint64_t> indptr(nrow + 1, 0);
    for (uint32_t r : tr) indptr[r + 1]++;
    for (size_t i = 0; i < nrow; i++) indptr[i + 1] += indptr[i];
    std::vector<uint64_t> fill(indptr.begin(), indptr.end() - 1);
    std::vector<std::pair<uint32_t, uint32_t>> ent(tr.size());
    for (size_t i = 0; i < tr.size(); i++) ent[fill[tr[i]]++] = {tc[i], tv[i]};
    std::vector<uint64_t> optr(nrow + 1, 0);
    std::vector<uint32_t> idx, val;
    idx.reserve(ent.size());
    val.reserve(ent.size());
    for (size_t r = 0; r < nrow; r++) {
        std::sort(ent.begin() + indptr[r], ent.begin() + indptr[r + 1]);
        for (uint64_t p = indptr[r]; p < indptr[r + 1]; p++) {
            if (!idx.empty() && idx.size() > optr[r] && idx.back() == ent[p].first)
                val.back() += ent[p].second;
            else {
                idx.push_back(ent[p].first);
                val.push_back(ent[p].second);
            }
        }
        optr[r + 1] = idx.size();
    }
    return AdaptiveMat::from_csmat(nrow, ncol, Storage::CSR, optr.data(), idx.data(), val.data());
}

// array_to_csv (tools/src/bin/cmd.rs:91-104): gz, comma separated, `{}` formatting of f64 (shortest
// round-trip decimal, never exponent notation)
static void array_to_csv(const double *a, size_t rows, size_t cols, const std::string &path) {
    gzFile f = gzopen(path.c_str(), "wb");
    if (!f) throw Error(SCANRS_ERR_ARGUMENT, path);
    char buf[512];
    for (size_t r = 0; r < rows; r++) {
        for (size_t c = 0; c < cols; c++) {
            auto res = std::to_chars(buf, buf + sizeof(buf), a[r * cols + c], std::chars_format::fixed);
            gzwrite(f, buf, (unsigned)(res.ptr - buf));
            if (c + 1 < cols) gzwrite(f, ",", 1);
        }
        gzwrite(f, "\n", 1);
    }
    gzclose(f);
}

int main(int argc, char **argv) {
    std::string input, out_dir = ".", norm = "cellranger";
    size_t num_pcs = 10;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string {
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", a.c_str());
                exit(2);
            }
            return argv[++i];
        };
        if (a == "-o" || a == "--out_dir")
            out_dir = next();
        else if (a == "-n" || a == "--norm")
            norm = next();
        else if (a == "-d" || a == "--num_pcs")
            num_pcs = strtoull(next().c_str(), nullptr, 10);
        else if (a == "-h" || a == "--help") {
            printf("scan-rs-cmd INPUT -o OUT_DIR -n NORMALIZATION -d NUM_PCS\n");
            return 0;
        } else
            input = a;
    }
    if (input.empty()) {
        fprintf(stderr, "error: the following required arguments were not provided: <INPUT>\n");
        return 2;
    }
    try {
        const Normalization normalization = normalization_from_str(norm);
        // extension over cmd.rs: a 10x feature-barcode .h5 is read the way diff-exp/src/utils.rs:42 reads it
        // (hdf5_io::matrix::read_adaptive_csr_matrix, all feature types, no count filter)
        const bool is_h5 = input.size() > 3 && input.compare(input.size() - 3, 3, ".h5") == 0;
        AdaptiveMat matrix = is_h5 ? hdf5_io::read_adaptive_csr_matrix(input).to_device() : load_mtx(input);
        mkdir(out_dir.c_str(), 0777);
        AdaptiveMat norm_mat = normalize(matrix.view(), normalization); // cmd.rs:67-80
        const PcaResult r = BkSvd().run_pca(norm_mat, num_pcs);
        array_to_csv(r.u.data.data(), r.u.rows, r.u.cols, out_dir + "/svd_u.csv.gz");
        array_to_csv(r.s.data(), num_pcs, 1, out_dir + "/svd_d.csv.gz");
        array_to_csv(r.v.data.data(), r.v.rows, r.v.cols, out_dir + "/svd_v.csv.gz");
    } catch (const Error &e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
