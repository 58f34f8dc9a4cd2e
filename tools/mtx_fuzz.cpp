// ASan/UBSan fuzz of scanrs_mtx_read (host-only): g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Iinclude -Iscan-rs_amd/csrc tools/mtx_fuzz.cpp \
//   scan-rs_amd/csrc/mtx_reader.cpp scan-rs_amd/csrc/h5_matrix.cpp scan-rs_amd/csrc/h5lite.cpp -lz -lpthread -o /tmp/mtx_fuzz && /tmp/mtx_fuzz 20000
// Character-level mutations of a small file (and byte flips of the gzip stream): every call returns a status. Round 2: 20 000 mutants clean.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include <zlib.h>
#include "common_err.hpp"
static char g_err[1024];
namespace scanrs {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
void fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); throw Failure{code}; }
}
extern "C" const char *scanrs_last_error(void) { return g_err; }
int main(int argc, char **argv) {
    std::mt19937_64 g(7);
    std::string base = "%%MatrixMarket matrix coordinate integer general\n% c\n5 7 6\n1 1 3\n2 5 1\n5 7 9\n3 3 2\n1 1 4\n4 2 8\n";
    const char alphabet[] = "0123456789 \n\t%-+.e";
    long ok = 0, bad = 0;
    for (int it = 0; it < atoi(argv[1]); it++) {
        std::string s = base;
        int nm = 1 + g() % 4;
        for (int k = 0; k < nm; k++) {
            size_t pos = g() % s.size();
            switch (g() % 3) {
            case 0: s[pos] = alphabet[g() % (sizeof(alphabet) - 1)]; break;
            case 1: s.insert(pos, 1, alphabet[g() % (sizeof(alphabet) - 1)]); break;
            default: s.erase(pos, 1 + g() % 3);
            }
        }
        gzFile f = gzopen("/tmp/scanrs_mtx_fuzz_cur.mtx.gz", "wb"); gzwrite(f, s.data(), (unsigned)s.size()); gzclose(f);
        if (g() % 20 == 0) { FILE *t = fopen("/tmp/scanrs_mtx_fuzz_cur.mtx.gz", "r+b"); fseek(t, 0, SEEK_END); long n = ftell(t); if (n > 12) { fseek(t, g() % n, SEEK_SET); fputc((int)(g() & 0xFF), t); } fclose(t); }
        scanrs_h5_matrix *m = nullptr;
        int rc = scanrs_mtx_read("/tmp/scanrs_mtx_fuzz_cur.mtx.gz", &m);
        if (rc == 0) { ok++; scanrs_h5_matrix_free(m); } else bad++;
    }
    printf("ok %ld refused %ld\n", ok, bad);
}
