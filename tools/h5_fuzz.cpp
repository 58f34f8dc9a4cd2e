// Sanitizer fuzz of the HDF5 reader (host-only code, so AddressSanitizer / UBSan run on the CPU build):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Iinclude -Iscan-rs_amd/csrc tools/h5_fuzz.cpp \
//       scan-rs_amd/csrc/h5lite.cpp scan-rs_amd/csrc/h5_matrix.cpp -lz -o /tmp/h5_fuzz
//   /tmp/h5_fuzz 1500 tests/golden/*.h5
// Every iteration either overwrites 1-3 random bytes of a fixture (and truncates one file in ten) or — structure-aware, every
// second iteration — finds a dataspace message, a chunked-layout message, a chunk B-tree node, a group B-tree node or a
// fixed / extensible array header in the bytes and rewrites one of ITS fields with an adversarial value (extents and chunk
// extents that wrap 64-bit products, chunk offsets off the grid or past the dataspace, child pointers that point back at the
// node, filter masks, entry counts, page bits); then runs every reader entry point on it: each call must return a status,
// never touch memory it does not own, and must return promptly (a watchdog alarm aborts a hang). Round 2: 21 000 mutants over the seven
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
#include <unistd.h>
#include <csignal>
#include "common_err.hpp"
static char g_err[1024];
namespace scanrs {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
void fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); throw Failure{code}; }
}
extern "C" const char *scanrs_last_error(void) { return g_err; }
static const uint64_t EVIL[] = {0ull, 1ull, 7ull, 8ull, 2000ull, 0x7FFFFFFFull, 0x80000000ull, 0xFFFFFFFFull, (1ull << 61) + 1024ull, 1ull << 62, 1ull << 63,
                                ~0ull, ~0ull - 7ull, (1ull << 40), 65536ull, 3ull};
static void put(std::vector<char> &b, size_t pos, uint64_t v, unsigned n) {
    for (unsigned i = 0; i < n && pos + i < b.size(); i++) b[pos + i] = (char)(v >> (8 * i));
}
// one structure-aware mutation; returns false when the file has none of the structures
static bool structured(std::vector<char> &b, std::mt19937_64 &g) {
    struct Hit { size_t pos; int kind; };
    std::vector<Hit> hits;
    for (size_t i = 0; i + 64 < b.size(); i++) {
        const unsigned char *p = (const unsigned char *)b.data() + i;
        if (p[0] == 1 && p[1] >= 1 && p[1] <= 3 && p[2] <= 1 && p[3] == 0 && p[4] == 0 && p[5] == 0 && p[6] == 0 && p[7] == 0) hits.push_back({i, 0}); // dataspace v1
        if (p[0] == 2 && p[1] >= 1 && p[1] <= 3 && p[2] <= 1 && p[3] == 1) hits.push_back({i, 1});                                        // dataspace v2
        if (p[0] == 3 && p[1] == 2 && p[2] >= 2 && p[2] <= 4) hits.push_back({i, 2});                                                      // layout v3 chunked
        if (p[0] == 4 && p[1] == 2 && p[3] >= 2 && p[3] <= 4 && p[4] >= 1 && p[4] <= 8) hits.push_back({i, 3});                             // layout v4 chunked
        if (!memcmp(p, "TREE", 4)) hits.push_back({i, p[4] == 1 ? 4 : 5});
        if (!memcmp(p, "FAHD", 4)) hits.push_back({i, 6});
        if (!memcmp(p, "EAHD", 4)) hits.push_back({i, 7});
        if (!memcmp(p, "OHDR", 4) || !memcmp(p, "OCHK", 4)) hits.push_back({i, 8});
    }
    if (hits.empty()) return false;
    const Hit h = hits[g() % hits.size()];
    const uint64_t evil = EVIL[g() % (sizeof EVIL / sizeof *EVIL)];
    const unsigned char *p = (const unsigned char *)b.data() + h.pos;
    switch (h.kind) {
    case 0: put(b, h.pos + 8 + 8 * (g() % (2 * p[1])), evil, 8); break;            // an extent or a maximum extent
    case 1: put(b, h.pos + 4 + 8 * (g() % (2 * p[1])), evil, 8); break;
    case 2:                                                                        // B-tree address or a chunk extent
        if (g() % 3 == 0) put(b, h.pos + 3, g() % 2 ? evil : (uint64_t)(g() % b.size()), 8);
        else put(b, h.pos + 11 + 4 * (g() % p[2]), evil, 4);
        break;
    case 3: put(b, h.pos + 5 + (size_t)p[4] * (g() % p[3]), evil, p[4]); break;     // a chunk extent (enc bytes)
    case 4: {                                                                      // chunk B-tree node: entries used, a key field, a child
        const unsigned r = (unsigned)(g() % 6);
        if (r == 0) put(b, h.pos + 6, g() % 2 ? evil : 1 + g() % 64, 2);
        else if (r == 1) put(b, h.pos + 5, g() % 4, 1);                             // level
        else {
            const size_t ent = h.pos + 24 + (g() % 4) * (8 + 8 * 3 + 8);            // rank-2 keys: size, mask, 3 offsets, child
            const unsigned f = (unsigned)(g() % 6);
            if (f == 0) put(b, ent, evil, 4);
            else if (f == 1) put(b, ent + 4, evil, 4);
            else if (f <= 4) put(b, ent + 8 + 8 * (f - 2), evil, 8);
            else put(b, ent + 32, g() % 2 ? (uint64_t)h.pos : evil, 8);             // child = this node: a cycle
        }
        break;
    }
    case 5:                                                                        // group B-tree node
        if (g() % 2) put(b, h.pos + 6, 1 + g() % 64, 2);
        else put(b, h.pos + 24 + 8 + (g() % 4) * 16, g() % 2 ? (uint64_t)h.pos : evil, 8);
        break;
    case 6: put(b, h.pos + 4 + g() % 12, evil, g() % 2 ? 1 : 8); break;             // client, entry size, page bits, element count
    case 7: put(b, h.pos + 4 + g() % 16, evil, g() % 2 ? 1 : 8); break;
    default: put(b, h.pos + 4 + g() % 24, evil, 1 + g() % 4); break;
    }
    return true;
}
int main(int argc, char **argv) {
    std::vector<std::string> files(argv + 2, argv + argc);
    int iters = atoi(argv[1]);
    std::mt19937_64 g(1);
    signal(SIGALRM, [](int) { const char m[] = "h5_fuzz: a reader call did not return within 20 s (hang)\n"; (void)!write(2, m, sizeof m - 1); _exit(3); });
    long ok = 0, bad = 0;
    for (auto &fn : files) {
        std::ifstream in(fn, std::ios::binary); std::vector<char> raw((std::istreambuf_iterator<char>(in)), {});
        for (int it = 0; it < iters; it++) {
            std::vector<char> b = raw;
            alarm(20);
            if (!(it % 2 == 1 && structured(b, g))) {
                int nflip = 1 + g() % 3;
                for (int k = 0; k < nflip; k++) { size_t pos = g() % b.size(); b[pos] = (char)(g() & 0xFF); }
                if (g() % 10 == 0) b.resize(g() % b.size() + 1);
            } else if (g() % 4 == 0) {
                structured(b, g); // two fields at once
            }
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
