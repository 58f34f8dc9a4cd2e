/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * Plain-C CPU restatement of the serial sparse loops of the scan-rs reference
 * (10XGenomics/scan-rs) for the normalize -> PCA hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).  Compile with -ffp-contract=off: the reference
 * computes `*o + *r * lval` as a separate multiply and add (rustc never
 * contracts to FMA without fast-math), see sqz/src/prod.rs:143-146.
 *
 * Parity pin: checked against the reference's inline known-answer tables
 * (tests/golden/*.json, transcribed from normalization.rs:560-721,
 * sqz/src/mat.rs:1282-1370, matrix_map.rs:420-446, stats.rs:73-81).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* ---- MatrixMap programs (sqz/src/matrix_map.rs) -------------------------
 * A lazily composed map chain (ComposedMap :145-197) is flattened into an op
 * list evaluated inner-to-outer per nonzero, exactly as ComposedMap::map
 * :189-192 nests the calls. */
enum {
    OP_INTO = 0,          /* MatrixIntoMap :98-130 (u32 -> f64)                   */
    OP_SCALE_AXIS = 1,    /* ScaleAxis :221-257: axis 0 -> f[r]*v, axis 1 -> f[c]*v */
    OP_LN_1P = 2,         /* ScalarMap |x| (x + 1.0).ln()   normalization.rs:173   */
    OP_LOG2_1P = 3,       /* ScalarMap |x| (x + 1.0).log2() normalization.rs:174   */
    OP_LOG10_1P = 4,      /* ScalarMap |x| (x + 1.0).log10() normalization.rs:175  */
    OP_SQUARE = 5,        /* ScalarMap |x| x.powi(2)  sqz/src/mat.rs:995           */
    OP_BINOM_DEV = 6,     /* BinomDevMap  normalization.rs:273-301                 */
    OP_BINOM_PEARSON = 7  /* BinomPearsonMap normalization.rs:332-351              */
};

typedef struct {
    int32_t kind;
    int32_t axis;
    int32_t swap; /* 1: this op sits under an odd number of TransposeMap wrappers (matrix_map.rs:72-75) */
    int32_t _pad;
    const double *a; /* scale factors, or n[c] for the binomial maps */
    const double *b; /* pi[r] for the binomial maps */
} oracle_op;

/* normalization.rs:263-269 */
static double a_ln_a_over_b(double a, double b) {
    if (a == 0.0) return 0.0;
    return a * log(a / b);
}

static double signum(double x) {
    /* f64::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN */
    if (isnan(x)) return x;
    return signbit(x) ? -1.0 : 1.0;
}

/* op->swap implements TransposeMap::map (matrix_map.rs:72-75): the wrapped map sees (c, r). */
static inline double eval_map(const oracle_op *ops, int n_ops, uint32_t v, size_t r0, size_t c0) {
    double x = (double)v;
    for (int i = 0; i < n_ops; i++) {
        const oracle_op *op = &ops[i];
        size_t r = op->swap ? c0 : r0;
        size_t c = op->swap ? r0 : c0;
        switch (op->kind) {
        case OP_INTO:
            break;
        case OP_SCALE_AXIS:
            x = (op->axis == 0 ? op->a[r] : op->a[c]) * x;
            break;
        case OP_LN_1P:
            x = log(x + 1.0);
            break;
        case OP_LOG2_1P:
            x = log2(x + 1.0);
            break;
        case OP_LOG10_1P:
            x = log10(x + 1.0);
            break;
        case OP_SQUARE:
            x = x * x;
            break;
        case OP_BINOM_DEV: {
            double n = op->a[c], pi = op->b[r];
            double mu = n * pi;
            double sign = signum(x - mu);
            double inner = 2.0 * (a_ln_a_over_b(x, mu) + a_ln_a_over_b(n - x, n - mu));
            /* f64::max(NaN-aware): .max(0.0) */
            double residual = sign * sqrt(fmax(inner, 0.0));
            double zero_term = -(sqrt(2.0 * n * log(1.0 / (1.0 - pi))));
            x = residual - zero_term;
            break;
        }
        case OP_BINOM_PEARSON: {
            double n = op->a[c], pi = op->b[r];
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

/* Evaluate the map on every stored nonzero (to_csmat value order, sqz/src/mat.rs:207-240). */
void oracle_map_values(int storage_csc, size_t n_outer, const uint64_t *indptr, const uint32_t *indices,
                       const uint32_t *values, const oracle_op *ops, int n_ops, double *out) {
    for (size_t o = 0; o < n_outer; o++) {
        for (uint64_t p = indptr[o]; p < indptr[o + 1]; p++) {
            size_t r = storage_csc ? indices[p] : o;
            size_t c = storage_csc ? o : indices[p];
            out[p] = eval_map(ops, n_ops, values[p], r, c);
        }
    }
}

/* sqz/src/prod.rs:30-51 + :123-148 (CSR) and :56-81 + :190-214 (CSC).
 * out must be zero-initialised by the caller (AdaptiveMat::dot, mat.rs:1084-1089).
 * Stored zeros are skipped as AbsIter::next does (sqz/src/vec.rs:113). */
void oracle_spmm_f64(int storage_csc, size_t n_outer, const uint64_t *indptr, const uint32_t *indices,
                     const uint32_t *values, const oracle_op *ops, int n_ops, const double *rhs,
                     size_t l, double *out) {
    if (!storage_csc) {
        for (size_t row = 0; row < n_outer; row++) {
            double *o = out + row * l;
            for (uint64_t p = indptr[row]; p < indptr[row + 1]; p++) {
                if (values[p] == 0) continue;
                size_t ind = indices[p];
                double lval = eval_map(ops, n_ops, values[p], row, ind);
                const double *r = rhs + ind * l;
                for (size_t j = 0; j < l; j++) o[j] = o[j] + r[j] * lval;
            }
        }
    } else {
        for (size_t col = 0; col < n_outer; col++) {
            const double *r = rhs + col * l;
            for (uint64_t p = indptr[col]; p < indptr[col + 1]; p++) {
                if (values[p] == 0) continue;
                size_t ind = indices[p];
                double lval = eval_map(ops, n_ops, values[p], ind, col);
                double *o = out + ind * l;
                for (size_t j = 0; j < l; j++) o[j] = o[j] + r[j] * lval;
            }
        }
    }
}

/* Same loops with A = u32 (the generic `A: Num` instantiation exercised by
 * sqz/src/mat.rs:1406-1486 and sqz/benches/my_benchmark.rs); identity map.
 * Unsigned arithmetic wraps (rustc release profile). */
void oracle_spmm_u32(int storage_csc, size_t n_outer, const uint64_t *indptr, const uint32_t *indices,
                     const uint32_t *values, const uint32_t *rhs, size_t l, uint32_t *out) {
    if (!storage_csc) {
        for (size_t row = 0; row < n_outer; row++) {
            uint32_t *o = out + row * l;
            for (uint64_t p = indptr[row]; p < indptr[row + 1]; p++) {
                if (values[p] == 0) continue;
                const uint32_t *r = rhs + (size_t)indices[p] * l;
                uint32_t lval = values[p];
                for (size_t j = 0; j < l; j++) o[j] = o[j] + r[j] * lval;
            }
        }
    } else {
        for (size_t col = 0; col < n_outer; col++) {
            const uint32_t *r = rhs + col * l;
            for (uint64_t p = indptr[col]; p < indptr[col + 1]; p++) {
                if (values[p] == 0) continue;
                uint32_t *o = out + (size_t)indices[p] * l;
                uint32_t lval = values[p];
                for (size_t j = 0; j < l; j++) o[j] = o[j] + r[j] * lval;
            }
        }
    }
}

/* sum_axis, sqz/src/mat.rs:377-406: axis 0 -> per-column sums, axis 1 -> per-row sums,
 * accumulated in storage order.  out zero-initialised by the caller. */
void oracle_sum_axis_f64(int storage_csc, size_t n_outer, const uint64_t *indptr, const uint32_t *indices,
                         const uint32_t *values, const oracle_op *ops, int n_ops, int axis,
                         double *out) {
    for (size_t o = 0; o < n_outer; o++) {
        for (uint64_t p = indptr[o]; p < indptr[o + 1]; p++) {
            if (values[p] == 0) continue;
            size_t r = storage_csc ? indices[p] : o;
            size_t c = storage_csc ? o : indices[p];
            double v = eval_map(ops, n_ops, values[p], r, c);
            out[axis == 0 ? c : r] += v;
        }
    }
}

/* sum_axis::<u32> on the raw count matrix (normalization.rs:159,161). */
void oracle_sum_axis_u32(int storage_csc, size_t n_outer, const uint64_t *indptr, const uint32_t *indices,
                         const uint32_t *values, int axis, uint32_t *out) {
    for (size_t o = 0; o < n_outer; o++) {
        for (uint64_t p = indptr[o]; p < indptr[o + 1]; p++) {
            size_t r = storage_csc ? indices[p] : o;
            size_t c = storage_csc ? o : indices[p];
            out[axis == 0 ? c : r] += values[p];
        }
    }
}

/* mean_var_axis accumulation, sqz/src/mat.rs:285-330 (sums only; the caller finishes
 * mean = s/m, var = s2/m - mean^2 as lines :323-327 do). */
void oracle_sum_sq_axis_f64(int storage_csc, size_t n_outer, const uint64_t *indptr, const uint32_t *indices,
                            const uint32_t *values, const oracle_op *ops, int n_ops, int axis,
                            double *sum, double *sumsq) {
    for (size_t o = 0; o < n_outer; o++) {
        for (uint64_t p = indptr[o]; p < indptr[o + 1]; p++) {
            if (values[p] == 0) continue;
            size_t r = storage_csc ? indices[p] : o;
            size_t c = storage_csc ? o : indices[p];
            double v = eval_map(ops, n_ops, values[p], r, c);
            size_t k = axis == 0 ? c : r;
            sum[k] += v;
            sumsq[k] += v * v; /* .powi(2) */
        }
    }
}
