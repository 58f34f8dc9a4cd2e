// Error plumbing shared by every translation unit, including the host-only ones (h5lite.cpp, h5_matrix.cpp) that are
// built without the HIP headers.
#pragma once
#include "scanrs_amd.h"

namespace scanrs {

// ---- errors: thrown inside, turned into status codes at the C boundary ----------
struct Failure {
    int code;
};
void set_error(const char *fmt, ...);
[[noreturn]] void fail(int code, const char *fmt, ...);

// ---- process-wide options of the entry points that take no handle (scanrs_set_global_option) ----------
struct GlobalOptions {
    int h5_threads = 8;                       // threads that inflate the chunks of a large filtered HDF5 read
    int eig_threads = 4;                      // host team of the tridiagonalisation for Rayleigh-Ritz matrices of 768+ rows (1, 2 or 4)
    int knn_exhaustive = 0;                   // 1: never use the bf16-MFMA filter of the k nearest neighbour search
    unsigned long long knn_filter_min_points = 32768; // point sets below this are ranked exhaustively
    unsigned long long knn_ratio = 4;         // density ratio between two subsets of the filter
    int knn_stats = 0;                        // 1: filter statistics on stderr
};
GlobalOptions &global_options();

} // namespace scanrs
