// Device-side pieces shared by the sparse kernels (kernels.hip, quad.hip): vector typedefs, the MatrixMap evaluation
// (the flattened ComposedMap chain of sqz/src/matrix_map.rs:189-192), wave helpers. gfx950 only.
#pragma once
#include "common.hpp"

namespace scanrs {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// MatrixMap evaluation: the flattened ComposedMap chain (sqz/src/matrix_map.rs:189-192), one
// nonzero per lane. `outer`/`inner` are positions in the copy being walked.
__device__ __forceinline__ double a_ln_a_over_b(double a, double b) { return a == 0.0 ? 0.0 : a * log(a / b); }

// Logarithms of the normalisation maps (`(x + 1.0).ln() / .log2() / .log10()`, scan-rs/src/normalization.rs:172-176).
// One log per nonzero per pass is most of the VALU work of the scan-like kernels, and ocml's f64 log is ~100 VALU
// instructions (double-double arithmetic for < 1 ulp). This is the fdlibm e_log.c kernel — argument reduced to
// m in [sqrt(1/2), sqrt(2)), s = f/(2+f), degree-14 odd polynomial, error < 2^-58 before the last roundings — with the
// division done by v_rcp_f64 + two Newton steps + one residual correction: ~45 instructions, result within 2 ulp of
// the correctly rounded value (checked against numpy in tests/test_gpu_parity.py). Arguments outside
// [1e-300, 1e300] (zero, negative, inf, NaN, subnormal) take the library routine.
__device__ __forceinline__ double log_core(double x, double &kd) {
    int e;
    double m = frexp(x, &e); // [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    kd = (double)e;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double s = f * r;
    s = fma(fma(-d, s, f), r, s);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 =
        z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return f - (hfsq - s * (hfsq + R)); // log(m)
}
__device__ __forceinline__ bool log_fast_range(double x) { return x >= 1e-300 && x <= 1e300; }
__device__ __forceinline__ double map_log2(double x) {
    if (!log_fast_range(x)) return log2(x);
    double kd;
    const double lm = log_core(x, kd);
    return fma(lm, 1.44269504088896338700e+00, kd);
}
__device__ __forceinline__ double map_ln(double x) {
    if (!log_fast_range(x)) return log(x);
    double kd;
    const double lm = log_core(x, kd);
    return fma(kd, 6.93147180369123816490e-01, lm + kd * 1.90821492927058770002e-10); // k ln2_hi + (log m + k ln2_lo)
}
__device__ __forceinline__ double map_log10(double x) {
    if (!log_fast_range(x)) return log10(x);
    double kd;
    const double lm = log_core(x, kd);
    // k log10(2) + log(m) / ln(10), log10(2) split so that k * hi is exact
    return fma(kd, 3.01029995663611771306e-01, fma(lm, 4.34294481903251816668e-01, kd * 3.69423907715893078616e-13));
}

constexpr int SCAN_U = 4; // strides of 64 nonzeros in flight per trip of the scan-like passes (SpMV, sums, moments)

// the chain from link `start` on, applied to x
__device__ __forceinline__ double eval_map_from(const DevMap &m, int start, double x, uint32_t outer, uint32_t inner) {
    for (int i = start; i < m.n; i++) {
        const DevOp &op = m.ops[i];
        switch (op.kind) {
        case OP_SCALE_AXIS:
            x = op.a[op.a_outer ? outer : inner] * x;
            break;
        case OP_LN_1P:
            x = map_ln(x + 1.0);
            break;
        case OP_LOG2_1P:
            x = map_log2(x + 1.0);
            break;
        case OP_LOG10_1P:
            x = map_log10(x + 1.0);
            break;
        case OP_SQUARE:
            x = x * x;
            break;
        case OP_BINOM_DEV: { // BinomDevMap::map, scan-rs/src/normalization.rs:279-296
            double n = op.a[op.a_outer ? outer : inner], pi = op.b[op.b_outer ? outer : inner];
            double mu = n * pi;
            double d = x - mu;
            double sign = (d != d) ? d : (signbit(d) ? -1.0 : 1.0);
            double inner2 = 2.0 * (a_ln_a_over_b(x, mu) + a_ln_a_over_b(n - x, n - mu));
            double residual = sign * sqrt(fmax(inner2, 0.0));
            double zero_term = -(sqrt(2.0 * n * log(1.0 / (1.0 - pi))));
            x = residual - zero_term;
            break;
        }
        case OP_BINOM_PEARSON: { // BinomPearsonMap::map, normalization.rs:338-347
            double n = op.a[op.a_outer ? outer : inner], pi = op.b[op.b_outer ? outer : inner];
            double mu = n * pi;
            double residual = (x - mu) / sqrt(mu * (1.0 - pi));
            double zero_term = -sqrt(n * pi / (1.0 - pi));
            x = residual - zero_term;
            break;
        }
        default:
            break;
        }
    }
    return x;
}
__device__ __forceinline__ double eval_map(const DevMap &m, uint32_t v, uint32_t outer, uint32_t inner) {
    return eval_map_from(m, 0, (double)v, outer, inner);
}
// the same chain for maps made of ScaleAxis / log / square links only (every normalisation but the binomial residuals):
// without the residual maps' square roots and library logarithms the evaluation needs far fewer registers, which the
// register-bound LDS-staged product (quad.hip) depends on
__device__ __forceinline__ double eval_map_simple_from(const DevMap &m, int start, double x, uint32_t outer, uint32_t inner) {
    for (int i = start; i < m.n; i++) {
        const DevOp &op = m.ops[i];
        switch (op.kind) {
        case OP_SCALE_AXIS:
            x = op.a[op.a_outer ? outer : inner] * x;
            break;
        case OP_LN_1P:
            x = map_ln(x + 1.0);
            break;
        case OP_LOG2_1P:
            x = map_log2(x + 1.0);
            break;
        case OP_LOG10_1P:
            x = map_log10(x + 1.0);
            break;
        case OP_SQUARE:
            x = x * x;
            break;
        default:
            break;
        }
    }
    return x;
}
inline bool map_is_simple(const DevMap &m) {
    for (int i = 0; i < m.n; i++)
        if (m.ops[i].kind == OP_BINOM_DEV || m.ops[i].kind == OP_BINOM_PEARSON) return false;
    return true;
}
// A chain that STARTS with a ScaleAxis indexed by the outer position (the barcode scale while walking a cell's vector)
// multiplies every count of the vector by the same number: it is read once per vector instead of once per nonzero per
// link evaluation (same product, same rounding).
struct RowMap {
    int start;
    double pre;
};
__device__ __forceinline__ RowMap row_map(const DevMap &m, uint32_t outer) {
    RowMap r;
    r.start = 0;
    r.pre = 1.0;
    if (m.n > 0 && m.ops[0].kind == OP_SCALE_AXIS && m.ops[0].a_outer) {
        r.start = 1;
        r.pre = m.ops[0].a[outer];
    }
    return r;
}
__device__ __forceinline__ double eval_map(const DevMap &m, const RowMap &rm, uint32_t v, uint32_t outer, uint32_t inner) {
    return eval_map_from(m, rm.start, rm.start ? rm.pre * (double)v : (double)v, outer, inner);
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}

// acc += w[lane n of this lane's row of 16] * x: the broadcast of the weight happens inside the instruction (DPP), where the
// round-3 form spent two v_readlane per general position. Nothing in front of it may have written `w` (2 wait states) or
// EXEC (5) with a vector instruction, and every lane must be switched on (a lane whose source lane is off is not written).
// tiles.hip: the weights come from a load, nothing in the position loop writes EXEC. The hazard recognizer does not look into inline
// assembly, so the two wait states travel with the instruction (s_nop 1: a register copy the allocator might place in front of it
// is then harmless; its issue slot hides in the software pipeline). The dense kernel's stream (tile_dense_body.inc) is generated:
// its weights are written by ds_bpermute_b32 and waited for with lgkmcnt, no vector instruction writes them.
__device__ __forceinline__ double fmac_bcast(double acc, double w, double x, int n) {
#define SCANRS_FB(N) \
    case N: asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x)); break;
    switch (n) {
        SCANRS_FB(0) SCANRS_FB(1) SCANRS_FB(2) SCANRS_FB(3) SCANRS_FB(4) SCANRS_FB(5) SCANRS_FB(6) SCANRS_FB(7)
        SCANRS_FB(8) SCANRS_FB(9) SCANRS_FB(10) SCANRS_FB(11) SCANRS_FB(12) SCANRS_FB(13) SCANRS_FB(14) SCANRS_FB(15)
    }
#undef SCANRS_FB
    return acc;
}



} // namespace scanrs
