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
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* All-core variant for bench.py's "generous" CPU baseline (1 core = the reference, which is
 * single-threaded with sequential MKL; all cores = what a maintainer could get from rayon). 1 (default) keeps every loop
 * below exactly the serial reference order — the parity pins are taken in that mode. With T > 1 threads the outer
 * vectors are dealt to the threads; loops that scatter (output indexed by the INNER position) accumulate into
 * per-thread copies of the output that are summed in thread order afterwards, so sums are re-associated. */
static int g_threads = 1;
void oracle_set_threads(int t) { g_threads = t < 1 ? 1 : t; }
int oracle_get_threads(void) { return g_threads; }
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

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
#ifdef _OPENMP
    if (g_threads > 1) {
        if (!storage_csc) {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 64)
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
            /* scatter: the number of output rows is the largest inner index + 1 */
            size_t n_inner = 0;
            for (uint64_t p = 0; p < indptr[n_outer]; p++)
                if ((size_t)indices[p] + 1 > n_inner) n_inner = (size_t)indices[p] + 1;
            const int T = g_threads;
            double *priv = (double *)calloc((size_t)T * n_inner * l, sizeof(double));
#pragma omp parallel num_threads(T)
            {
                double *mine = priv + (size_t)omp_get_thread_num() * n_inner * l;
#pragma omp for schedule(static)
                for (size_t col = 0; col < n_outer; col++) {
                    const double *r = rhs + col * l;
                    for (uint64_t p = indptr[col]; p < indptr[col + 1]; p++) {
                        if (values[p] == 0) continue;
                        size_t ind = indices[p];
                        double lval = eval_map(ops, n_ops, values[p], ind, col);
                        double *o = mine + ind * l;
                        for (size_t j = 0; j < l; j++) o[j] = o[j] + r[j] * lval;
                    }
                }
#pragma omp for schedule(static)
                for (size_t e = 0; e < n_inner * l; e++) {
                    double acc = out[e];
                    for (int t = 0; t < T; t++) acc += priv[(size_t)t * n_inner * l + e];
                    out[e] = acc;
                }
            }
            free(priv);
        }
        return;
    }
#endif
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

void oracle_sum_sq_axis_f64(int storage_csc, size_t n_outer, const uint64_t *indptr, const uint32_t *indices,
                            const uint32_t *values, const oracle_op *ops, int n_ops, int axis, double *sum, double *sumsq);

/* sum_axis, sqz/src/mat.rs:377-406: axis 0 -> per-column sums, axis 1 -> per-row sums,
 * accumulated in storage order.  out zero-initialised by the caller. */
void oracle_sum_axis_f64(int storage_csc, size_t n_outer, const uint64_t *indptr, const uint32_t *indices,
                         const uint32_t *values, const oracle_op *ops, int n_ops, int axis,
                         double *out) {
#ifdef _OPENMP
    if (g_threads > 1) {
        oracle_sum_sq_axis_f64(storage_csc, n_outer, indptr, indices, values, ops, n_ops, axis, out, NULL);
        return;
    }
#endif
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
#ifdef _OPENMP
    if (g_threads > 1) {
        const int outer_axis = storage_csc ? 0 : 1; /* the axis whose index is the outer vector */
        if (axis == outer_axis) {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 256)
            for (size_t o = 0; o < n_outer; o++) {
                double s1 = 0.0, s2 = 0.0;
                for (uint64_t p = indptr[o]; p < indptr[o + 1]; p++) {
                    if (values[p] == 0) continue;
                    size_t r = storage_csc ? indices[p] : o;
                    size_t c = storage_csc ? o : indices[p];
                    double v = eval_map(ops, n_ops, values[p], r, c);
                    s1 += v;
                    s2 += v * v;
                }
                sum[o] += s1;
                if (sumsq) sumsq[o] += s2;
            }
        } else {
            size_t n_inner = 0;
            for (uint64_t p = 0; p < indptr[n_outer]; p++)
                if ((size_t)indices[p] + 1 > n_inner) n_inner = (size_t)indices[p] + 1;
            const int T = g_threads;
            double *priv = (double *)calloc((size_t)T * n_inner * 2, sizeof(double));
#pragma omp parallel num_threads(T)
            {
                double *m1 = priv + (size_t)omp_get_thread_num() * n_inner * 2, *m2 = m1 + n_inner;
#pragma omp for schedule(static)
                for (size_t o = 0; o < n_outer; o++) {
                    for (uint64_t p = indptr[o]; p < indptr[o + 1]; p++) {
                        if (values[p] == 0) continue;
                        size_t r = storage_csc ? indices[p] : o;
                        size_t c = storage_csc ? o : indices[p];
                        double v = eval_map(ops, n_ops, values[p], r, c);
                        m1[indices[p]] += v;
                        m2[indices[p]] += v * v;
                    }
                }
#pragma omp for schedule(static)
                for (size_t e = 0; e < n_inner; e++) {
                    double a1 = sum[e], a2 = sumsq ? sumsq[e] : 0.0;
                    for (int t = 0; t < T; t++) {
                        a1 += priv[(size_t)t * n_inner * 2 + e];
                        a2 += priv[(size_t)t * n_inner * 2 + n_inner + e];
                    }
                    sum[e] = a1;
                    if (sumsq) sumsq[e] = a2;
                }
            }
            free(priv);
        }
        return;
    }
#endif
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
