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

} // namespace scanrs
