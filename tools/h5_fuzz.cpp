// Sanitizer fuzz of the HDF5 reader (host-only code, so AddressSanitizer / UBSan run on the CPU build):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Iinclude -Iscan-rs_amd/csrc tools/h5_fuzz.cpp \
//       scan-rs_amd/csrc/h5lite.cpp scan-rs_amd/csrc/h5_matrix.cpp -lz -o /tmp/h5_fuzz
//   /tmp/h5_fuzz 1500 tests/golden/*.h5
// Every iteration overwrites 1-3 random bytes of a fixture (and truncates one file in ten), then runs every reader entry
// point on it: each call must return a status, never touch memory it does not own. Round 2: 21 000 mutants over the seven
// fixtures, clean (it found two real bugs on the way: an indptr entry past nnz dereferenced before it was validated, and
// feature id / type tables of different lengths indexed by the same counter).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>
#include <random>
#include "common_err.hpp"
static char g_err[1024];
namespace scanrs {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
void fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); throw Failure{code}; }
}
extern "C" const char *scanrs_last_error(void) { return g_err; }
int main(int argc, char **argv) {
    std::vector<std::string> files(argv + 2, argv + argc);
    int iters = atoi(argv[1]);
    std::mt19937_64 g(1);
    long ok = 0, bad = 0;
    for (auto &fn : files) {
        std::ifstream in(fn, std::ios::binary); std::vector<char> raw((std::istreambuf_iterator<char>(in)), {});
        for (int it = 0; it < iters; it++) {
            std::vector<char> b = raw;
            int nflip = 1 + g() % 3;
            for (int k = 0; k < nflip; k++) { size_t pos = g() % b.size(); b[pos] = (char)(g() & 0xFF); }
            if (g() % 10 == 0) b.resize(g() % b.size() + 1);
            { std::ofstream o("/tmp/scanrs_h5_fuzz_cur.h5", std::ios::binary); o.write(b.data(), b.size()); }
            scanrs_h5_matrix *m = nullptr;
            int rc = scanrs_h5_read_csc_matrix("/tmp/scanrs_h5_fuzz_cur.h5", &m);
            if (rc == 0) { ok++; scanrs_h5_matrix_free(m); } else bad++;
            rc = scanrs_h5_read_adaptive_csr_matrix("/tmp/scanrs_h5_fuzz_cur.h5", "Gene", 1, &m);
            if (rc == 0) scanrs_h5_matrix_free(m);
            uint64_t n = 0, nb = 0, dims[8]; uint32_t rank;
            std::vector<double> out(100000);
            for (const char *ds : {"f64_2d_edge", "u64_be_chunked", "u32_many_chunks", "i32_compact", "fixed_array_paged", "fixed_array_filtered_2d", "implicit", "ea/ea_small", "ea/ea_filtered", "ea/ea_super", "ea/ea_2d_unlim0", "ea/ea_2d_unlim1", "clustering/_graphclust/clusters", "all_differential_expression/_graphclust/data"})
                scanrs_h5_read_f64("/tmp/scanrs_h5_fuzz_cur.h5", ds, out.data(), out.size(), dims, &rank);
            std::vector<char> buf(100000);
            scanrs_h5_member_names("/tmp/scanrs_h5_fuzz_cur.h5", "many", buf.data(), buf.size(), &n, &nb);
            scanrs_h5_read_strings("/tmp/scanrs_h5_fuzz_cur.h5", "strings", buf.data(), buf.size(), &n, &nb);
        }
    }
    printf("ok %ld refused %ld\n", ok, bad);
}
