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
        AdaptiveMat matrix = is_h5 ? hdf5_io::read_adaptive_csr_matrix(input).to_device() : mtx::load_mtx(input); // mtx.rs:10-51, in the library
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
