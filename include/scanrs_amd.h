/*
 * scanrs_amd.h — C ABI of the MI355X (gfx950) implementation of scan-rs's
 * sparse-count-matrix normalize -> PCA hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, opaque handle,
 * create -> operate -> free, exactly the style the reference itself uses for its
 * one native component (bhtsne/src/bindings.rs:8-31).  Each entry point names
 * the reference interface it replaces (paths relative to 10XGenomics/scan-rs).
 * INTEGRATION.md shows the Rust `extern "C"` block and the `DataMat` / `Dot` /
 * `Pca` impls a maintainer would add on the reference side.
 *
 * Conventions
 *  - Matrices keep the reference orientation: `rows x cols`, `storage` 0 = CSR,
 *    1 = CSC (same u8 code as sqz/src/mat.rs:45-65). Cell Ranger's matrix is
 *    features x barcodes.
 *  - Dense panels are row-major, standard layout (sqz/src/prod.rs:102-106), f64
 *    unless the name says u32.
 *  - All functions return a status (0 = ok); they never unwind.  The message
 *    for the calling thread's last failure is scanrs_last_error().
 *  - Host pointers are borrowed for the duration of the call; the library owns
 *    the device copies inside the handle.  A handle is not thread-safe;
 *    distinct handles are independent.
 *  - There is no CPU fallback: every compute entry point needs a gfx950 device
 *    and fails with SCANRS_ERR_DEVICE otherwise.
 */
#ifndef SCANRS_AMD_H
#define SCANRS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scanrs_mat scanrs_mat; /* opaque: AdaptiveMat / LowRankOffset on the device */

/* status codes (anyhow errors / panics / CancellationError of the reference) */
enum {
    SCANRS_OK = 0,
    SCANRS_ERR_SHAPE = 1,     /* "The input matrix must be at least 2x2." / "Dimension mismatch" */
    SCANRS_ERR_INVALID_K = 2, /* "invalid k" (bk_svd.rs:77-79) */
    SCANRS_ERR_CANCELLED = 3, /* snoop::CancellationError (snoop/src/lib.rs:5-18) */
    SCANRS_ERR_DEVICE = 4,    /* no gfx950 device / HIP failure / out of memory */
    SCANRS_ERR_NUMERICAL = 5, /* LAPACK-style failure (`?` on qr()/svddc_into()) */
    SCANRS_ERR_ARGUMENT = 6,  /* null pointer, bad enum, unsupported combination */
    SCANRS_ERR_IO = 7         /* file missing / not HDF5 / truncated / a format feature this reader does not parse */
};

/* storage flag, sqz/src/mat.rs:45-65 */
enum { SCANRS_CSR = 0, SCANRS_CSC = 1 };

/* scalar maps (`ScalarMap`, sqz/src/matrix_map.rs:269-308; closures of normalization.rs:172-176, mat.rs:995) */
enum { SCANRS_FN_LN_1P = 2, SCANRS_FN_LOG2_1P = 3, SCANRS_FN_LOG10_1P = 4, SCANRS_FN_SQUARE = 5 };

/* `enum Normalization`, scan-rs/src/normalization.rs:11-28 (same order) */
enum {
    SCANRS_NORM_CELLRANGER = 0,
    SCANRS_NORM_CELLRANGER8 = 1,
    SCANRS_NORM_SEURATLOG = 2,
    SCANRS_NORM_BINOMIAL_DEVIANCE = 3,
    SCANRS_NORM_BINOMIAL_PEARSON = 4,
    SCANRS_NORM_WITH_SIZE_FACTORS = 5,
    SCANRS_NORM_LOG_TRANSFORM = 6
};

const char *scanrs_last_error(void);
/* 1 if a gfx950 device is usable from this process, else 0 (never fails). */
int scanrs_device_available(void);
const char *scanrs_version(void);

/* ---- storage: sqz::AdaptiveMat (sqz/src/mat.rs:34-42) ---------------------- */

/* AdaptiveMat::from_csmat (mat.rs:92-124) / hdf5-io read_adaptive_csr_matrix
 * (hdf5-io/src/matrix.rs:119-192): take a compressed matrix by its
 * indptr(u64)/indices(u32)/data(u32) triplet (host pointers), upload it.
 * Indices must be ascending within each outer vector; stored zeros are dropped
 * (AbsIter semantics, sqz/src/vec.rs:113). The map is MatrixIntoMap. */
int scanrs_mat_create(uint64_t rows, uint64_t cols, int storage, const uint64_t *indptr, const uint32_t *indices,
                      const uint32_t *values, scanrs_mat **out);
/* The reader's fallback for 10x matrix files whose indices are not sorted within a cell
 * (`new_from_unsorted_csc`, hdf5-io/src/matrix.rs:66-75): same as scanrs_mat_create, but every outer vector is first
 * sorted by index on the device. Repeated indices inside one vector are still an error. */
int scanrs_mat_create_unsorted(uint64_t rows, uint64_t cols, int storage, const uint64_t *indptr, const uint32_t *indices,
                               const uint32_t *values, scanrs_mat **out);
/* Same, the triplet already lives in device memory (copied, not adopted). */
int scanrs_mat_create_device(uint64_t rows, uint64_t cols, int storage, const uint64_t *d_indptr,
                             const uint32_t *d_indices, const uint32_t *d_values, scanrs_mat **out);

/* One sqz::AdaptiveVec as it lies in memory (sqz/src/vec.rs:1029-1053): the encoded buffers are handed over
 * untouched and decoded on the device (AbsIter::next, vec.rs:96-117: ascending positions, stored zeros skipped).
 *   kind      0 D3, 1 D4, 2 D8, 3 D16, 4 V, 5 S3, 6 S4, 7 S8  (declaration order of `enum AdaptiveVec`)
 *   len       logical length (the matrix's inner dimension)
 *   n_units   D*: = len; V: stored entries; S*: stored entries (= length of the inner dense value vector)
 *   data      D3/S3: Dense3.data (u64 words, 21 fields each, vec.rs:895-900); D4/S4: Dense4.data (two nibbles per
 *             byte, low nibble = even position, vec.rs:761-766); D8/S8, D16: DenseW.data (vec.rs:660-664); V: NULL
 *   fallback_*  the SimpleSparse fallback of the dense vector (indexes ascending; keyed by position for D*, by
 *             entry number for S*); for V the vector's own indexes / values (vec.rs:123-127)
 *   index_bytes, block_starts   CompressedIndexSparse fields (vec.rs:222-227), S* only:
 *             block_starts has round_up(len,256)/256 + 1 entries */
typedef struct scanrs_adaptive_vec {
    uint32_t kind;
    uint64_t len;
    uint64_t n_units;
    const void *data;
    uint64_t data_bytes;
    const uint32_t *fallback_indexes;
    const uint32_t *fallback_values;
    uint64_t n_fallback;
    const uint8_t *index_bytes;
    const uint32_t *block_starts;
    uint64_t n_block_starts;
} scanrs_adaptive_vec;

/* AdaptiveMat::new(rows, cols, storage, Vec<AdaptiveVec>) (sqz/src/mat.rs:68-90): `n_vecs` must be the outer
 * dimension (rows for CSR, cols for CSC) and every vector `len` the inner one. The compressed buffers
 * (about 4 kB per cell) are uploaded and expanded on the device; the handle is the same as scanrs_mat_create's. */
int scanrs_mat_create_adaptive(uint64_t rows, uint64_t cols, int storage, const scanrs_adaptive_vec *vecs, uint64_t n_vecs,
                               scanrs_mat **out);
/* Drop for AdaptiveMat / LowRankOffset (Rust `Drop`, cf. bhtsne/src/lib.rs:19-23). Null is a no-op. */
void scanrs_mat_free(scanrs_mat *m);

/* AdaptiveMat::view (mat.rs:242-245): new handle sharing the storage, same map/offset. */
int scanrs_mat_view(const scanrs_mat *m, scanrs_mat **out);
/* AdaptiveMat::t / LowRankOffset::t (mat.rs:262-270, low_rank_offset.rs:60-65): transposed view. */
int scanrs_mat_t(const scanrs_mat *m, scanrs_mat **out);

/* rows(), cols(), nnz(), storage (mat.rs:155-180) */
int scanrs_mat_shape(const scanrs_mat *m, uint64_t *rows, uint64_t *cols);
int scanrs_mat_nnz(const scanrs_mat *m, uint64_t *nnz);
int scanrs_mat_storage(const scanrs_mat *m, int *storage);

/* ---- lazy maps: sqz::MatrixMap (sqz/src/matrix_map.rs) ----------------------- */

/* set_map(MatrixIntoMap) (mat.rs:892-898): back to the raw counts; also drops the offset. */
int scanrs_mat_reset_map(scanrs_mat *m);
/* compose_map(ScaleAxis::new(Axis(axis), factors)) (mat.rs:901-913, matrix_map.rs:221-257).
 * axis 0: factors[r] * v (length rows); axis 1: factors[c] * v (length cols). */
int scanrs_mat_compose_scale_axis(scanrs_mat *m, int axis, const double *factors);
/* apply(f) = compose_map(ScalarMap::new(f)) (mat.rs:925-933); f is one of SCANRS_FN_*. */
int scanrs_mat_apply(scanrs_mat *m, int scalar_fn);
/* LowRankOffset::new(mat, u, v) (low_rank_offset.rs:26-33): u is rows x rank, v is rank x cols. */
int scanrs_mat_set_offset(scanrs_mat *m, uint32_t rank, const double *u, const double *v);

/* center / scale / scale_and_center (mat.rs:937-1001). `given` may be null
 * (means / std-devs are then computed on the device as the reference does). */
int scanrs_mat_center(scanrs_mat *m, int axis, const double *given_means);
int scanrs_mat_scale(scanrs_mat *m, int axis, const double *given_std);
int scanrs_mat_scale_and_center(scanrs_mat *m, int axis, const double *given_scaling);

/* ---- reductions (mat.rs:273-406) ------------------------------------------------ */

/* sum_axis::<u32> on the raw counts (normalization.rs:159,161). axis 0 -> cols entries. */
int scanrs_mat_sum_axis_u32(scanrs_mat *m, int axis, uint32_t *out);
/* sum_axis::<f64> of the mapped values (offset not included, as in the reference). */
int scanrs_mat_sum_axis_f64(scanrs_mat *m, int axis, double *out);
int scanrs_mat_mean_axis(scanrs_mat *m, int axis, double *out);
int scanrs_mat_mean_var_axis(scanrs_mat *m, int axis, double *mean, double *var);

/* to_dense (mat.rs:188-205, low_rank_offset.rs:55-57): rows x cols f64, small matrices / tests. */
int scanrs_mat_to_dense(scanrs_mat *m, double *out);

/* ---- products: `Dot` impls (mat.rs:1074-1170, low_rank_offset.rs:68-96, prod.rs) -- */

/* self.dot(rhs): rhs is cols x l, out is rows x l (includes the offset u*(v*rhs) when set). */
int scanrs_mat_dot(scanrs_mat *m, const double *rhs, uint32_t l, double *out);
/* lhs.dot(self): lhs is l x rows, out is l x cols. */
int scanrs_mat_rdot(scanrs_mat *m, const double *lhs, uint32_t l, double *out);
/* The `A = u32` instantiation tested by mat.rs:1406-1486 / benched by
 * sqz/benches/my_benchmark.rs (identity map, wrapping integer arithmetic). */
int scanrs_mat_dot_u32(scanrs_mat *m, const uint32_t *rhs, uint32_t l, uint32_t *out);
int scanrs_mat_rdot_u32(scanrs_mat *m, const uint32_t *lhs, uint32_t l, uint32_t *out);
/* Device-resident panels, padded leading dimensions (elements): d_out[rows x l] = self * d_rhs[cols x l];
 * with `transpose` != 0: d_out[cols x l] = self^T * d_rhs[rows x l]. ld must be even. */
int scanrs_mat_dot_device(scanrs_mat *m, int transpose, const double *d_rhs, uint32_t ld_rhs, uint32_t l, double *d_out,
                          uint32_t ld_out);

/* ---- normalization: scan-rs/src/normalization.rs ----------------------------------- */

/* normalize / normalize_with_size_factor (normalization.rs:46-102) and the binomial
 * residual maps (:232-322): installs the lazy map + rank-1 offset on the handle.
 * size_factors (length cols) only for SCANRS_NORM_WITH_SIZE_FACTORS, else null. */
int scanrs_normalize(scanrs_mat *m, int normalization, const uint32_t *size_factors);
/* log_normalize_with_size_factor (normalization.rs:138-178) without the centre/scale step.
 * umi_count_sum < 0 means None (median of the column sums). */
int scanrs_log_normalize(scanrs_mat *m, double umi_count_sum, int log_fn, const uint32_t *size_factors);
/* log1p_normalize_fixed_point (normalization.rs:191-213). */
int scanrs_log1p_normalize_fixed_point(scanrs_mat *m, int log_fn, uint32_t base, uint32_t exponent);
/* target UMI count chosen by the last (log_)normalize call: max(median(colsums), 1) (normalization.rs:162-168) */
int scanrs_mat_target_umi(const scanrs_mat *m, double *target);

/* ---- PCA: scan-rs/src/dim_red ----------------------------------------------------------- */

/* snoop::CancelProgress (snoop/src/lib.rs:37-58): `cancel` points at an
 * AtomicBool-compatible byte read with relaxed ordering between kernel launches
 * (may be null); `progress` receives set_progress fractions (may be null). */
typedef void (*scanrs_progress_fn)(void *ctx, double fraction);
typedef struct {
    const volatile uint8_t *cancel;
    scanrs_progress_fn progress;
    void *ctx;
} scanrs_snoop;

/* BkSvd::run_pca_cancellable / svd_bk (dim_red/bk_svd.rs:41-146).
 * omega: optional explicit start panel in the reference's own layout
 * ((cols x b) row-major when rows >= cols, else (b x rows)), b = min(rows, cols, ceil(k*k_multiplier));
 * null -> generated from `seed` like SmallRng::seed_from_u64 + Uniform(-1,1).
 * Outputs (caller-allocated, row-major): u rows x k, s k, v cols x k  (= PcaResult, dim_red/mod.rs:47).
 * u and/or v may be null: the factor then stays in device memory only (no PCIe copy). */
int scanrs_pca_bk(scanrs_mat *m, uint32_t k, double k_multiplier, uint32_t n_iter, uint64_t seed, const double *omega,
                  const scanrs_snoop *snoop, double *u, double *s, double *v);
/* RandSvd::run_pca / svd_rand (dim_red/rand_svd.rs:37-129). l = max(k+4, (k*l_multiplier) as usize).
 * omega layout: (cols x l) when rows >= cols, else (l x rows). */
int scanrs_pca_rand(scanrs_mat *m, uint32_t k, double l_multiplier, uint32_t n_iter, uint64_t seed, const double *omega,
                    double *u, double *s, double *v);
/* Irlba::run_pca_cancellable / irlba (dim_red/irlba.rs:59-215). v0: optional start vector (length cols);
 * null -> seeded normal. mprod (may be null) receives the "number of matrix products" (irlba.rs:212). */
int scanrs_pca_irlba(scanrs_mat *m, uint32_t nu, double tol, uint32_t max_iter, const double *v0,
                     const scanrs_snoop *snoop, double *u, double *s, double *v, uint32_t *mprod);
/* PcaResult of the last scanrs_pca_bk / scanrs_pca_rand call on this handle as it lies in device memory
 * (dim_red/mod.rs:47 returns owned arrays; a device consumer — scanrs_knn_device below, a clustering kernel of the
 * caller — takes them from here without the PCIe round trip): *d_u is rows x k with leading dimension *ld_u
 * (elements, row-major), *d_v is cols x k (the local columns of a sharded handle). The memory belongs to the handle
 * and is valid until its next PCA call or scanrs_mat_free. Any output pointer may be null. */
int scanrs_pca_result_device(scanrs_mat *m, const double **d_u, uint32_t *ld_u, const double **d_v, uint32_t *ld_v,
                             uint32_t *k);
/* The panel the two randomized drivers draw for a given seed (count values, row-major fill order). */
int scanrs_omega_fill(uint64_t seed, uint64_t count, double *out);

/* ---- nearest neighbours of the PCA scores: scan_rs::nn (scan-rs/src/nn.rs) -------------- */

/* knn(v, k) (nn.rs:38-57): for every row of the n x d row-major matrix `points` the indices of its k nearest OTHER rows
 * by Euclidean distance, nearest first; rows of `out` (n x k) are padded with UINT32_MAX when fewer than k exist
 * (`T::max_value()`, nn.rs:66). Exact (exhaustive search in f64 on the device); exactly equidistant neighbours come in
 * ascending index order. k <= 128, d <= 128. */
int scanrs_knn(const double *points, uint64_t n, uint32_t d, uint32_t k, uint32_t *out);
/* find_nn(v, k, tree, include_self) (nn.rs:63-83): the k nearest of `points` (the tree's point set, n_p x d) to each
 * row of `queries` (n_q x d). As in the reference, include_self = 0 drops the point whose INDEX equals the query's
 * row number, which is only meaningful when the two sets are the same. */
int scanrs_find_nn(const double *queries, uint64_t n_q, const double *points, uint64_t n_p, uint32_t d, uint32_t k,
                   int include_self, uint32_t *out);

/* scanrs_knn on points already in device memory (n x d, row-major, leading dimension ld >= d elements), e.g. the
 * scores scanrs_pca_result_device hands out; `out` (n x k) is a host array. */
int scanrs_knn_device(const double *d_points, uint64_t n, uint32_t ld, uint32_t d, uint32_t k, uint32_t *out);

/* ---- multi-GPU: one process per GPU, cells range-partitioned ----------- */

/* In-place sum all-reduce of `count` elements of device memory across ranks.
 * dtype 0 = f64, 1 = u64. Returns 0 on success. The host program supplies it
 * (bench.py: torch.distributed over RCCL); the library calls it once per
 * sparse product that contracts over the sharded dimension. */
typedef int (*scanrs_allreduce_fn)(void *ctx, void *d_buf, uint64_t count, int dtype);
/* Declare that this handle holds outer vectors [outer_begin, outer_begin + n_local) of a
 * matrix whose sharded (outer) dimension has `outer_global` entries. The handle's own
 * shape keeps the local count; reductions over the outer dimension become global. */
int scanrs_mat_set_shard(scanrs_mat *m, uint32_t rank, uint32_t world, uint64_t outer_begin, uint64_t outer_global,
                         scanrs_allreduce_fn allreduce, void *ctx);
/* The library's own transport for the exchange steps (replaces the host hook above): RCCL over xGMI, loaded when the
 * first communicator is made. One process per GPU: rank 0 draws the 128-byte id (scanrs_comm_get_unique_id), the
 * host program passes it to the other ranks by any channel it has, every rank calls scanrs_comm_create with its
 * device current, then scanrs_mat_set_shard_comm on its handle. The collectives are enqueued on the handle's own
 * stream (no host synchronisation). The communicator is borrowed by the handle and must outlive it. */
typedef struct scanrs_comm scanrs_comm;
#define SCANRS_COMM_ID_BYTES 128
int scanrs_comm_get_unique_id(uint8_t *id /* SCANRS_COMM_ID_BYTES */);
int scanrs_comm_create(const uint8_t *id, uint32_t rank, uint32_t world, scanrs_comm **out);
void scanrs_comm_free(scanrs_comm *c);
/* The group as the transport counts it - ranks and this rank from ncclCommCount / ncclCommUserRank (the single-process form: its own
 * group) - and the sum all-reduces that went through this communicator so far (calls, payload bytes). Any pointer may be NULL. */
int scanrs_comm_info(scanrs_comm *c, uint32_t *nranks, uint32_t *rank, uint64_t *n_allreduce, uint64_t *allreduce_bytes);
int scanrs_mat_set_shard_comm(scanrs_mat *m, scanrs_comm *comm, uint32_t rank, uint32_t world, uint64_t outer_begin,
                              uint64_t outer_global);

/* Single-process form (SURVEY.md §8b `mat_create(..., n_gpus)`; Cell Ranger is one process, tools/src/bin/cmd.rs:61-70):
 * the whole matrix is handed over once, its outer vectors are range-partitioned by nonzeros over `n_shards` devices
 * (`devices` null = 0 .. n_shards-1; ids may repeat, several shards then share a device), every shard is driven by a
 * host thread of the library during a call, and the exchange steps are a one-shot reduce-scatter + all-gather over
 * peer-mapped memory. Outputs as in scanrs_pca_bk: u rows x k, s k, v cols x k for the WHOLE matrix. */
typedef struct scanrs_multi scanrs_multi;
int scanrs_multi_create(uint64_t rows, uint64_t cols, int storage, const uint64_t *indptr, const uint32_t *indices,
                        const uint32_t *values, uint32_t n_shards, const int *devices, scanrs_multi **out);
void scanrs_multi_free(scanrs_multi *mm);
int scanrs_multi_n_shards(const scanrs_multi *mm, uint32_t *n);
/* shard i: its handle (owned by mm; use it from a thread whose current device is *device), its range of outer vectors */
int scanrs_multi_shard(scanrs_multi *mm, uint32_t i, scanrs_mat **shard, int *device, uint64_t *outer_begin, uint64_t *outer_end);
int scanrs_multi_normalize(scanrs_multi *mm, int normalization, const uint32_t *size_factors);
int scanrs_multi_pca_bk(scanrs_multi *mm, uint32_t k, double k_multiplier, uint32_t n_iter, uint64_t seed, const double *omega,
                        const scanrs_snoop *snoop, double *u, double *s, double *v);
int scanrs_multi_pca_rand(scanrs_multi *mm, uint32_t k, double l_multiplier, uint32_t n_iter, uint64_t seed, const double *omega,
                          double *u, double *s, double *v);
int scanrs_multi_log_normalize(scanrs_multi *mm, double umi_count_sum, int log_fn, const uint32_t *size_factors);
/* v0 (optional) spans ALL columns of the matrix. */
int scanrs_multi_pca_irlba(scanrs_multi *mm, uint32_t nu, double tol, uint32_t max_iter, const double *v0, const scanrs_snoop *snoop,
                           double *u, double *s, double *v, uint32_t *mprod);
/* nnz-balanced contiguous partition of the outer dimension: bounds has world+1 entries. */
int scanrs_plan_shards(const uint64_t *indptr, uint64_t n_outer, uint32_t world, uint64_t *bounds);

/* ---- measurement ------------------------------------------------------------------------- */

/* Per-kernel-class HIP-event timing on the handle's stream (bench.py roofline leg). */
typedef struct {
    char name[48];
    uint64_t launches;
    double total_ms;
    double algorithmic_bytes;   /* summed over launches, SURVEY.md §8d accounting */
    double onchip_gather_bytes; /* panel bytes gathered through L2 / L1 by the sparse products (nnz * 8 * l), else 0 */
} scanrs_kernel_stat;
int scanrs_profile_enable(scanrs_mat *m, int on);
int scanrs_profile_reset(scanrs_mat *m);
/* Fills up to `cap` entries, writes the total count to *n. */
int scanrs_profile_get(scanrs_mat *m, scanrs_kernel_stat *out, uint32_t cap, uint32_t *n);
/* Which f64 product kernel serves this handle (and its views): 0 = auto (L2-blocked gather for large matrices and
 * panels of 16+ columns, plain gather otherwise), 1 = plain gather, 2 = L2-blocked gather, 3 = hybrid: LDS-staged panel
 * tiles over a tile-bucketed layout of the matrix plus the L2-blocked gather over the nonzeros that overflow it, run side
 * by side (panels of 16..104 columns; the layout is built on first use under a given map; other widths take path 2).
 * All are HIP kernels; results agree to rounding. */
int scanrs_mat_set_spmm_path(scanrs_mat *m, int path);
/* Tuning options of a handle (shared by its views); defaults are the measured optimum on MI355X.
 *   "tile_k" (2)        hybrid product: record positions per (outer vector, visit): 2, 3 or 4
 *   "tile_s" (32)       hybrid product: outer vectors per wave (32, or 28 with tile_k 2 or 4: 168 instead of 184 registers per tile wave)
 *   "ov_tile_kb" (0)    hybrid product: panel slice per step of the overflow gather (0 = twice l2_tile_kb)
 *   "tile_max_overflow" (0.35)  auto path: an orientation whose tile layout would leave more than this share of the nonzeros to
 *                       the overflow gather (dense outer vectors: genes detected in most cells) stays on the gather kernels
 *   "tile_ku" (1)       hybrid product, tile_k 2: 1 = one of the two positions takes only count-1 nonzeros, whose rows are added
 *                       without a weight (maps whose value at count 1 is an outer factor times an inner factor); 0 = none
 *   "tile_t" (48)       hybrid product: panel rows per tile (<= 24 tile_k)
 *   "tile_b" (4)        hybrid product: tile buffers in the LDS ring (tile_t * tile_b <= 192); a nonzero may wait tile_b - 2 visits
 *   "tile_auto" (1)     path 0 may use the hybrid product for matrices of 2^24+ nonzeros (layout built when svd_bk / svd_rand
 *                       start, or at the second product under the same map, if the device has room: ~12 B per nonzero)
 *   "tile_overlap" (1)  hybrid product: the overflow gather runs beside the tile kernel (0: after it; measurement only)
 *   "l2_tile_kb" (3584) panel slice per step of the L2-blocked gather
 *   "spmm_order" (1)    L2-blocked gather launch order: 0 storage order, 1 longest vectors first when a launch is a few
 *                       rounds of waves, 2 always longest first
 *   "hot_segment" (512) nonzeros per step from which a vector gets a whole workgroup (0 = never)
 *   "materialize" (1)   keep the values of the map prefix per nonzero on the copy with few, long outer vectors
 *   "slice_walk" (1)    Ix1 products / moments on the copy with few, long vectors stage the inner-indexed arrays in LDS slices
 *   "spmv_lds" (1)      Ix1 products on the copy with many short vectors stage the vector in LDS parts
 *   "overlap" (1)       small dense work of the solvers runs on a second stream beside the sparse passes
 *   "col_moments" (1)   mean_var_axis / sum of a log-normalized map (scale factors on the summed-over axis, then a logarithm):
 *                       walk the copy whose outer vectors are the summed-over axis — a table of the map at counts 1..8 per outer
 *                       vector instead of one logarithm per nonzero, sums scattered into LDS as 64-bit fixed point (order-
 *                       independent, so bit-reproducible). Taken when that copy exists and the matrix is large; 2: whenever
 *                       eligible; 0: never. Results agree with the ordinary pass to ~1e-13 relative.
 *   "device_factor" (1) svd_bk: the b x b Cholesky factors of CholeskyQR and the coefficient bookkeeping of qr(K) stay on the
 *                       device (no host round trip per orthonormalisation; falls back to the host path by itself when a
 *                       factorization does not converge within the queued passes); 0: host factorizations
 *   "d2h_threads" (4)   host threads that empty the pinned ring of a large result download
 *   "tile_spare_cus" (1) the persistent tile kernel launches as many workgroups as its number of item rounds needs (3 977 equal items:
 *                       16 rounds on 256 workgroups and on 249); the CUs left over serve the side streams during the pass (0: one per CU)
 *   "spmv_row_table" (1) Ix1 products over many short outer vectors (IRLBA's A v on the cell-major copy): a map that depends on the count
 *                       and the outer position alone is looked up by count from a table made per vector (0: values materialized per nonzero)
 *   "gemm_direct" (1)   dense panel products X W read their operands straight from memory into the MFMA registers
 *                       (0: the LDS-tiled kernels)
 *   "reuse_cmax" (1e5)  svd_bk: coefficient bound above which a projection column is recomputed directly
 *   "side_build" (1)    scanrs_normalize starts a helper thread (own stream) that builds what the PCA behind it needs: the first
 *                       product's tile layout, the transposed copy of the matrix and the second orientation's layout, in that
 *                       order, beside the normalisation passes and the solver's first product (a solver called without a
 *                       normalize before it starts the helper for the second orientation itself); 2: the helper starts behind
 *                       the normalisation passes instead of beside them (measured the same); 0: everything is built on demand by
 *                       the calling thread
 *   "tile_split" (1)    tile layout (default tile shape): an outer vector owns as many SLOTS (units of tile_k record positions per visit)
 *                       as its density asks for: V = round(x / tile_split_x), at least 1, at most 32, x = its expected nonzeros per
 *                       panel tile; its nonzeros are dealt to its slots round-robin; below tile_split_min it owns none and all of it
 *                       goes to the overflow part. Makes the layout fit real count matrices (a few thousand genes detected in most
 *                       cells, most genes in almost none). 0: one slot per vector
 *   "tile_split_x" (1.8), "tile_split_min" (0.5)   the two densities of that rule, in nonzeros per tile
 *   "tile_build_one_pass" (1)  the wave-level layout builder writes the records and collects the nonzeros without a position in ONE
 *                       walk over the matrix (temporaries of 8 bytes per nonzero, then a compaction); 0: a counting walk first
 *   "tile_weights_wide" (1)  the weight refresh of a unit-mode layout works four positions per thread with wide loads and stores
 *                       (0: one position per thread; same values)
 *   "dense_side_no_lds" (0)  experiment: dense kernels queued on the side streams use the register-only MFMA forms, which can run
 *                       beside the persistent tile kernel (measured slower: DESIGN.md section 9)
 *   "tile_dense" (1)    tile layout of the default shape: 1 = DENSE record streams — the records of a (wave, visit) are a packed list,
 *                       as long as the most loaded of the item's 8 waves needs for the tile that leaves the ring, and the accumulator of
 *                       a record is selected at run time (VGPR index mode): about 1.05 record positions per nonzero and no overflow
 *                       beyond the vectors too sparse to own a slot; 0 = the round-4 form, tile_k fixed positions per (slot, visit)
 *   "tile_sort_slots" (2)  dense tile layout: 2 = the slots, sorted by load, are DEALT over the waves' groups (rank k -> group k mod n, accumulator
 *                       k div n): every group holds one slot of every load stratum, so the 8 waves of an item carry the same number of records
 *                       per visit AND all items of a part move through the panel's tiles at the same pace (a tile stays in L2 between the
 *                       first and the last workgroup that stages it: gene-major pass 15.5 -> 13.7 ms); 1 = 32 consecutive ranks per group
 *                       (rounds 5-6: equal waves, but items of very different load); 0 = vector order
 *   "tile_one_walk" (1)  dense tile layout: 1 = built in ONE walk over the matrix (the (group, part) blocks of the tile-sorted record
 *                       list start at the prefix sums of their capacities, found by binary searches; counts above 15 leave through
 *                       a bounded list); 0 = a counting walk and a filling walk. The same layout bit for bit.
 *   "tile_fold" (1)     dense tile layout under a map whose count-1 value is a product of a per-row and a per-column factor (every
 *                       normalisation of the reference): 1 = the weight of a record position holds only the factor of the side that owns
 *                       the logarithm (with the count's ratio); the other side's factor rides in the staged panel (columns of the
 *                       product's inner side) or is applied when a vector's sums are collected (outer side) - one table lookup per
 *                       position in the weight refresh instead of two; 0 = both factors in every weight. Same products to rounding.
 *   "tile_wtab" (1)     dense tile layout under such a map (with tile_fold 1): 1 = the product kernel evaluates the map itself - it gathers a
 *                       record position's weight from a table of 16 entries by count per place of the layout (or per inner position),
 *                       addressed by the record's own count and slot / ring row; no per-position weight exists in memory and a normalize
 *                       rewrites only the table. Counts above 15 are served by the overflow part, where the chain is evaluated per
 *                       nonzero. 0 = one f64 weight per record position, rewritten by every normalize (the form every other map takes)
 *   "tile_flow" (0)     dense tile layout with tile_wtab 1: 1 = the FLOW form - tiles of 32 rows in a ring of 6 (the same 192 rows of LDS), one
 *                       record stream per wave cut into rounds of 64 positions whatever tiles they belong to (about 1.01 positions per
 *                       nonzero), no barrier: the ring is handed over at ticks inside the streams through counters in LDS. Same products to
 *                       rounding; measured on par with the round-5 form, kept as an option. A map that does not
 *                       separate, or more counts above 15 than the one-walk build's list holds, makes an orientation fall back to 0.
 *   "tile_big_list_cap" (0)  dense tile layout, one-walk build: entries of the list that carries the nonzeros with counts above 15 to the
 *                       overflow part (0: max(4 M, nnz / 64)); a matrix with more of them is built by the two-walk form instead.
 *   "tile_emit_staged" (1)  dense tile layout, diagnostic: 0 makes the emission of the record streams search its per-visit tables in global
 *                       memory instead of LDS - the form taken by itself when a part has more than 4 000 tiles. Same layout.
 *   "tile_builder" (1)  1: wave-level builder of the tile layout (default tile shape); 0: per-thread walk (reference form)
 *   "tile_build_waves" (0)   cap on the waves per CU of that builder (0: as many as fit)
 *   "sync_timeout_s" (120)  PROCESS-WIDE (same as scanrs_set_global_option): deadline of every host-side wait for the device
 * Unknown keys return SCANRS_ERR_ARGUMENT. The only environment variables the library reads are the diagnostics
 * SCANRS_TRACE and SCANRS_TRACE_EIG (phase timings on stderr). */
int scanrs_mat_set_option(scanrs_mat *m, const char *key, double value);
/* Event counters of the handle (diagnostics): "bk_host_retries" = svd_bk calls whose device-side factorizations did not
 * converge within the queued passes and that were run again with host factorizations. First-call accounting, host microseconds of
 * the calling thread since the handle was made: "t_layout_us" (tile layout builds), "t_side_wait_us" (waiting for the helper
 * thread of "side_build"), "t_start_panel_us", "t_delivery_us" (U, V to host arrays); process-wide: "t_alloc_us" / "alloc_calls"
 * (hipMalloc). The tile layouts of the handle (both orientations, summed): "tile_positions" = record positions the tile kernel works
 * per pair of passes, "tile_served_nonzeros" = nonzeros among them (the rest is padding), "tile_overflow_nonzeros" = nonzeros left
 * to the overflow gather. */
int scanrs_mat_get_counter(scanrs_mat *m, const char *key, uint64_t *value);
/* Process-wide options of the entry points that take no handle:
 *   "h5_threads" (8)               threads that inflate the chunks of a large filtered HDF5 read
 *   "eig_threads" (4)              host team of the Rayleigh-Ritz eigensolver for matrices of 768+ rows (1, 2 or 4)
 *   "knn_exhaustive" (0)           1: never use the bf16-MFMA filter of scanrs_knn*
 *   "knn_filter_min_points" (32768), "knn_ratio" (4), "knn_stats" (0)   tuning / statistics of that filter
 *   "sync_timeout_s" (120)         BOUNDED WAITS: no call of this library blocks on the device without a deadline. Every wait for
 *                                  a stream or an event (and every barrier between the shard threads of scanrs_multi_*) is a poll
 *                                  with this deadline in seconds; when it passes, the call returns SCANRS_ERR_DEVICE and
 *                                  scanrs_last_error() names the wait (function, file:line), the calling thread's last solver
 *                                  stages and which of the handle's streams (main / aux / aux2 / overflow) still had work; the
 *                                  same report and the whole stage ring go to stderr. Copies that were queued may still run: the
 *                                  host buffers given to the failing call (and the device's view of the handle) must not be
 *                                  reused or freed before the process exits. The handle must then be freed (its queued
 *                                  work never finished); start over in a fresh process — a process whose device stopped
 *                                  answering cannot be repaired from inside, and must not exec() another program either.
 *   "device_cache_fraction" (0.5)  device blocks of 1 MB and more that the library releases are kept for its next allocation of about
 *                                  their size, up to this share of the device's memory, instead of going back to the driver (VRAM
 *                                  that was just freed is scrubbed in the background; an allocation that lands on it waits:
 *                                  seconds for the tens of GB of a handle). 0: no cache. See scanrs_release_cached_memory. */
int scanrs_set_global_option(const char *key, double value);
/* Optional: loads the library's device code, starts the one-off host-side table computation of the seeded start panels, pins the
 * 72 MB host staging buffer a PCA's result delivery and start panel go through (kept in a process-wide pool across handles) and
 * touches the runtime paths the first call would otherwise initialise (~30-60 ms on MI355X): call it at program start to
 * keep that out of the first scanrs_mat_create / normalize / PCA. Everything works without it. */
int scanrs_init(void);
/* Optional: one allocation of `bytes` made now, on the calling thread's current device, from which the library carves its later
 * large buffers (about 70 bytes per nonzero for a matrix that goes through normalize + PCA with the hybrid product). A caller who
 * knows the size of the matrix before it is loaded takes the allocation's latency — on a device whose memory was just freed by
 * another process the driver is still scrubbing it, and an allocation waits for that — out of the first PCA. The reserve is an
 * arena: blocks carved from it return to it when the work queued before their release has run, neighbouring holes merge, and a
 * later request of any size is carved from them again; the reserve itself goes back to the driver with
 * scanrs_release_cached_memory once none of it is in use. */
int scanrs_reserve_device_memory(uint64_t bytes);
/* Gives the cached device blocks (see "device_cache_fraction") back to the driver / reports how much is cached on the calling
 * thread's current device. */
int scanrs_release_cached_memory(void);
int scanrs_cached_memory_bytes(uint64_t *bytes);
/* Device memory in the library's buffers right now, all handles of the process (blocks waiting in the cache not counted). */
int scanrs_device_memory_in_use(uint64_t *bytes);
/* Arithmetic of the large sparse products: 0 (default) = f64 throughout, the reference's arithmetic; 1 = opt-in fast
 * mode: the dense panel is rounded to f32 before it is gathered (half the on-chip bytes per nonzero), products and
 * sums stay f64. Singular values / loadings then agree with the f64 path to ~1e-7 relative, not to rounding. */
int scanrs_mat_set_panel_precision(scanrs_mat *m, int precision);
/* Block until all work queued on the handle's stream is done. */
int scanrs_mat_sync(scanrs_mat *m);

/* One pass of the factor step of the device-side CholeskyQR (`.qr()` of a b-wide panel, bk_svd.rs:94,98,123,127): g is the
 * n x n Gram matrix of the panel (row-major, n <= 128), `rows` the panel's row count (shift rule), `pass` the pass number
 * (from pass 1 on the step first tests max |g - I| < 5e-14 sqrt(n) and answers "converged"). rinv receives R^-1 with
 * g (+ shift I) = R^T R — or the identity when the step has nothing to apply. done: 1 converged or failed; status: 0 ok,
 * 1 Cholesky failed after 12 shifts, 2 non-finite input; err = max |g - I|; shift = diagonal shift used. For tests. */
int scanrs_mat_chol_rinv(scanrs_mat *m, const double *g, uint32_t n, uint64_t rows, int pass, double *rinv, int *done, int *status,
                         double *err, double *shift);

/* Diagnostics of the bounded waits (no device needed): runs the library's wait loop on an event that is never signalled and
 * returns SCANRS_ERR_DEVICE once `timeout_s` seconds have passed, with the report a real stuck wait leaves behind. */
int scanrs_debug_wait_never(double timeout_s);
/* ... and the barrier between the shard threads of the single-process multi-GPU form, entered by one thread of `world` alone */
int scanrs_debug_barrier_alone(uint32_t world);
/* ... and the bookkeeping of a reserve made by scanrs_reserve_device_memory (no device needed: the arena is driven on a made-up address
 * range): `rounds` random rounds of carving and giving back blocks; SCANRS_OK when no two live blocks ever overlapped, every block
 * stayed inside the range, neighbouring holes always merged and the arena was whole again at the end. */
int scanrs_debug_arena_selftest(uint32_t rounds, uint64_t seed);

/* ---- host-side dense helpers (no device needed; used by the solvers where the reference calls
 * LAPACK on k x k matrices, exposed so the CPU test-suite can check them) ------------------------- */
/* in place upper Cholesky G = R^T R (row-major n x n); SCANRS_ERR_NUMERICAL when G is not SPD */
int scanrs_host_chol_upper(double *g, int n);
/* in place inverse of an upper-triangular matrix */
int scanrs_host_inv_upper(double *r, int n);
/* symmetric eigen-decomposition, w descending, z[i*n + j] = component i of eigenvector j */
int scanrs_host_sym_eig(const double *a, int n, double *w, double *z);
/* the k leading eigenpairs only: w[0..k) descending, z row-major n x k */
int scanrs_host_sym_eig_topk(const double *a, int n, int k, double *w, double *z);

/* ---- 10x HDF5 ingestion (hdf5-io/src/matrix.rs, analysis.rs). Host-side; no device needed.
 * The files are parsed by the library's own reader (csrc/h5lite.cpp) — no libhdf5 dependency. Failures are
 * SCANRS_ERR_IO with the reason in scanrs_last_error(). ------------------------------------------------------------ */
typedef struct scanrs_h5_matrix scanrs_h5_matrix; /* GenericFeatureBarcodeMatrix / MatrixMetadata (scan-types/src/matrix.rs:8-15) */

/* `read_csc_matrix` (hdf5-io/src/matrix.rs:56-97): group "matrix" -> features x barcodes CSC, u64 indptr, u32 indices,
 * u32 values (stored values converted through f64 as the reference does, :247-257). Columns whose indices are not
 * ascending (some Cell Ranger 3 files) are sorted, the reference's `new_from_unsorted_csc` fallback (:71-79). */
int scanrs_h5_read_csc_matrix(const char *path, scanrs_h5_matrix **out);
/* `read_adaptive_csr_matrix` (:129-199): the same matrix feature-major (CSR) with features dropped when their
 * feature_type does not contain `retain_feature_like` (NULL: keep all) or their total count is below `shrink_row`
 * (< 0: None). The arrays are exactly what scanrs_mat_create(rows, cols, SCANRS_CSR, ...) takes. */
int scanrs_h5_read_adaptive_csr_matrix(const char *path, const char *retain_feature_like, int64_t shrink_row,
                                       scanrs_h5_matrix **out);
/* `read_matrix_metadata` (:17-54): barcodes, (filtered) feature ids / names / types and nnz; no matrix arrays. */
int scanrs_h5_read_matrix_metadata(const char *path, const char *retain_feature_like, scanrs_h5_matrix **out);
void scanrs_h5_matrix_free(scanrs_h5_matrix *m);

int scanrs_h5_matrix_shape(const scanrs_h5_matrix *m, uint64_t *rows, uint64_t *cols, uint64_t *nnz, int *storage);
/* borrowed pointers, valid until scanrs_h5_matrix_free; NULL for a metadata-only handle */
int scanrs_h5_matrix_arrays(const scanrs_h5_matrix *m, const uint64_t **indptr, const uint32_t **indices, const uint32_t **values);
/* what = 0 barcodes, 1 feature ids, 2 feature names, 3 feature types (per kept feature; the LabelClass of the reference
 * flattened, scan-types/src/label_class.rs:129-145), 4 name of the file */
int scanrs_h5_matrix_n_strings(const scanrs_h5_matrix *m, int what, uint64_t *n);
const char *scanrs_h5_matrix_string(const scanrs_h5_matrix *m, int what, uint64_t i);
/* indices (in the file's feature order) of the features that were filtered out — the BTreeSet the reference returns */
int scanrs_h5_matrix_removed(const scanrs_h5_matrix *m, const uint64_t **removed, uint64_t *n);

/* One call from a file to the device handle: `read_adaptive_csr_matrix` (or `load_mtx` when `path` does not end in
 * ".h5") followed by scanrs_mat_create on the CSR arrays. `meta` (optional) receives the host-side handle with the
 * barcodes / feature tables / removed set; free it with scanrs_h5_matrix_free. Needs a gfx950 device. */
int scanrs_mat_create_from_file(const char *path, const char *retain_feature_like, int64_t shrink_row, scanrs_mat **out,
                                scanrs_h5_matrix **meta);

/* `load_mtx` (scan-rs/src/mtx.rs:10-51): gzipped MatrixMarket coordinate file -> CSR arrays in the same handle type
 * (no string tables; scanrs_h5_matrix_arrays / _shape / _free apply). Comments '%', header "NROW NCOL NNZ", 1-based
 * "ROW COL VAL" triplets with u32 values, duplicates summed, indices ascending inside a row (TriMat::to_csr). */
int scanrs_mtx_read(const char *path, scanrs_h5_matrix **out);

/* `read_umi_counts_from_matrix` (:270-299): per-barcode sums of the stored values, read in blocks of 2000 columns */
int scanrs_h5_read_umi_counts(const char *path, uint32_t *out, uint64_t cap, uint64_t *n);

/* `get_clustering_keys` (analysis.rs:38-41): names under /clustering, NUL-separated into buf */
int scanrs_h5_get_clustering_keys(const char *path, char *buf, uint64_t cap, uint64_t *n_keys, uint64_t *bytes);
/* `get_clustering` (analysis.rs:5-20): i64 -> i16 / u16 by truncation, as the reference's `as` casts do */
int scanrs_h5_get_clustering(const char *path, const char *clustering_key, uint16_t *num_clusters, int16_t *clusters,
                             uint64_t cap, uint64_t *n);
/* `get_differential_expression` (analysis.rs:23-36): the rows x cols f64 table, row-major */
int scanrs_h5_get_differential_expression(const char *path, const char *clustering_key, double *out, uint64_t cap,
                                          uint64_t *rows, uint64_t *cols);

/* generic access used by the tests to check the parser against files written by libhdf5: numbers of any stored type
 * converted to f64 (dims gets up to 8 entries), fixed-length strings NUL-separated, link names NUL-separated */
int scanrs_h5_read_f64(const char *path, const char *dataset, double *out, uint64_t cap, uint64_t *dims, uint32_t *rank);
int scanrs_h5_read_strings(const char *path, const char *dataset, char *buf, uint64_t cap, uint64_t *n, uint64_t *bytes);
int scanrs_h5_member_names(const char *path, const char *group, char *buf, uint64_t cap, uint64_t *n, uint64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* SCANRS_AMD_H */
