// Shared declarations of the scanrs_amd library (host side). gfx950 only.
#pragma once
#include <atomic>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <cstdint>
#include <cstdarg>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include <hip/hip_runtime.h>

#include "scanrs_amd.h"
#include "common_err.hpp"

struct scanrs_comm; // comm.cpp

namespace scanrs {

#define SCANRS_HIP(expr)                                                                              \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            ::scanrs::fail(SCANRS_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                                       \
    } while (0)

// SCANRS_TRACE=1: wall-clock of host-side phases on stderr (diagnostics only)
bool trace_on();
// Where the calling thread stands: kept in a small per-thread ring (always; a timed-out wait names the last ones) and printed with
// SCANRS_TRACE=2 (a line on stderr, no synchronisation)
void stage_mark(const char *what, long a = 0, long b = 0);

// ---- bounded waits ----------------------------------------------------------------------------------------------------
// Every host-side wait for the device goes through these: a poll (hipStreamQuery / hipEventQuery, spinning for the first
// 200 us, then short sleeps) with a deadline — option "sync_timeout_s", default 120 s — instead of a blocking
// hipStreamSynchronize / hipEventSynchronize: a device that never signals turns into SCANRS_ERR_DEVICE naming the wait
// (function, file:line), the thread's last stage marks and which of the handle's streams are still busy, not into a call that
// never returns. The *_quiet forms are for destructors and unwinding (bounded too, nothing thrown).
struct Storage;
void wait_stream(hipStream_t s, const char *func, const char *file, int line);
void wait_event(hipEvent_t e, const char *func, const char *file, int line);
void wait_device(const char *func, const char *file, int line);
bool wait_stream_quiet(hipStream_t s) noexcept;
bool wait_event_quiet(hipEvent_t e) noexcept;
double sync_timeout_s();
void set_sync_timeout_s(double s);
// the handle whose streams a timed-out wait on this thread reports (set by the entry points that run device work)
struct CurrentHandle {
    const Storage *prev;
    // Every entry point that queues device work makes one: after a timed-out wait (device_lost) it refuses until the device has been seen
    // idle again. `waits_only`: an entry point that only waits (scanrs_mat_sync) - the way to get there.
    explicit CurrentHandle(const Storage *st, bool waits_only = false);
    ~CurrentHandle();
};
#define SCANRS_SYNC(stream) ::scanrs::wait_stream((stream), __PRETTY_FUNCTION__, __FILE__, __LINE__)
// A few bytes from the device whose landing place outlives a timed-out wait (copies into stack variables of a frame that a
// timeout unwinds would land in whatever lives there later): the copy goes into a slot of a process-wide pinned ring and is read from
// there behind the wait; when the wait gives up, the slot is simply not read.
void *landing_slot(size_t bytes); // capi.cpp: 8-byte aligned, from a pinned ring of 1 MB that is never unmapped
// Device -> host for arrays: through a pinned staging area that is never unmapped, in pieces of at most 8 MB, each piece waited for
// (bounded) before it is copied on into `dst`. If a wait gives up, the copy still in flight lands in the staging area — not in a
// vector or a caller's array whose frame has unwound meanwhile. `src_pitch` / `row_bytes` / `rows`: a 2-D copy into a
// compact destination.
void d2h_landed(void *dst, const void *dsrc, size_t bytes, hipStream_t s, const char *func, const char *file, int line);
void d2h_landed_2d(void *dst, const void *dsrc, size_t src_pitch, size_t row_bytes, size_t rows, hipStream_t s, const char *func, const char *file, int line);
#define SCANRS_D2H(dst, dsrc, bytes, stream) ::scanrs::d2h_landed((dst), (dsrc), (bytes), (stream), __PRETTY_FUNCTION__, __FILE__, __LINE__)
#define SCANRS_D2H_2D(dst, dsrc, pitch, row_bytes, rows, stream) ::scanrs::d2h_landed_2d((dst), (dsrc), (pitch), (row_bytes), (rows), (stream), __PRETTY_FUNCTION__, __FILE__, __LINE__)
template <typename T>
inline T d2h_value(const T *d, hipStream_t s, const char *func, const char *file, int line) {
    T *slot = static_cast<T *>(landing_slot(sizeof(T)));
    const hipError_t e = hipMemcpyAsync(slot, d, sizeof(T), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) ::scanrs::fail(SCANRS_ERR_DEVICE, "hipMemcpyAsync failed: %s (%s:%d)", hipGetErrorString(e), file, line);
    void wait_stream(hipStream_t s, const char *func, const char *file, int line);
    wait_stream(s, func, file, line);
    return *slot;
}
#define SCANRS_D2H_VALUE(dptr, stream) ::scanrs::d2h_value((dptr), (stream), __PRETTY_FUNCTION__, __FILE__, __LINE__)
#define SCANRS_SYNC_EVENT(ev) ::scanrs::wait_event((ev), __PRETTY_FUNCTION__, __FILE__, __LINE__)
struct Tick {
    const char *what;
    std::chrono::steady_clock::time_point t0;
    explicit Tick(const char *w) : what(w), t0(std::chrono::steady_clock::now()) {}
    ~Tick() {
        if (trace_on())
            fprintf(stderr, "[scanrs trace] %-28s %8.3f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// ---- device memory ------------------------------------------------------------------
// hipFree waits for the whole device (a buffer dropped while a 20 ms sparse pass runs holds the host for those 20 ms), and VRAM
// that was just freed is scrubbed in the background: an allocation right behind a large free waits for the scrubber (0.24 s
// per 8 GB, profiles/microbench/alloc_probe2). So buffers are never freed in the middle of a call: release() hands the pointer
// to a list, together with one event per stream of the releasing handle; a block whose events have completed joins the cache at the
// next allocation, and goes back to the driver only where a call ends (end of create / a solver / scanrs_mat_sync / scanrs_mat_free)
// or when an allocation fails for lack of memory, which then tries again.
void device_free_later(void *p, size_t bytes);
void device_free_flush() noexcept;                             // released blocks whose release events have completed
void device_free_flush_owner_gone(const void *owner) noexcept; // blocks of a handle that has just been destroyed
bool device_lost();                                            // a bounded wait gave up earlier: nothing is reused or freed, entry points fail fast
void *device_alloc(size_t bytes); // hipMalloc with the retry above; throws Failure(SCANRS_ERR_DEVICE)
void device_cache_release() noexcept;
void device_reserve(size_t bytes);
void device_cache_set_fraction(double f);
size_t device_cache_bytes();
size_t device_live_bytes();
size_t device_reserve_unused_bytes();
void library_warm_up(); // kernels.hip: loads every code object of the library (one empty launch per translation unit)
uint64_t device_alloc_us();
uint64_t device_alloc_calls();

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            n = o.n;
            o.p = nullptr;
            o.n = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    static constexpr size_t SLACK = 4096;
    void alloc(size_t count) {
        release();
        n = count;
        if (!count) return;
        // Slack behind every buffer: kernels that read a vector in aligned 16-byte chunks (tile_assign_wave_kernel) look up to two
        // chunks past the end of the last vector; the tile kernel loads the records and weights of the next two visits without
        // asking whether they exist (spmm_tile_body: 2 visits x 2 sets x 64 weights x 8 B behind the last group's last visit)
        p = static_cast<T *>(device_alloc(count * sizeof(T) + SLACK));
    }
    void ensure(size_t count) {
        if (count > n) alloc(count);
    }
    void release() {
        if (p) device_free_later(p, n * sizeof(T) + SLACK);
        p = nullptr;
        n = 0;
    }
};

// Grow-only named scratch buffers: nothing is hipMalloc'ed inside the solver loops after
// the first pass over them.
struct Scratch {
    std::map<std::string, DevBuf<char>> bufs;
    hipStream_t stream = nullptr; // zero-fills are ordered on the owning handle's stream
    template <typename T>
    T *get(const std::string &key, size_t count) {
        auto &b = bufs[key];
        size_t bytes = count * sizeof(T);
        if (bytes > b.n) {
            // (an eighth more, so that a slightly larger request does not reallocate - up to 32 MB (uncapped, the two 30 GB projection panels of a
            // PCA of a 3.75 M-cell shard carried 7.5 GB of it))
            b.alloc(bytes + std::min<size_t>(bytes / 8, (size_t)32 << 20) + 256);
            SCANRS_HIP(hipMemsetAsync(b.p, 0, b.n, stream)); // padding columns of panels start out as zeros
        }
        return reinterpret_cast<T *>(b.p);
    }
    std::set<std::string> filled; // keys whose (constant) contents are already on the device
    void clear() {
        bufs.clear();
        filled.clear();
    }
};

// ---- per-kernel-class timing (scanrs_profile_*) -------------------------------------
struct Profile {
    bool on = false;
    struct Rec {
        std::string name;
        hipEvent_t a, b;
        double bytes, onchip;
    };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    struct Stat {
        uint64_t launches = 0;
        double ms = 0, bytes = 0, onchip = 0;
    };
    std::map<std::string, Stat> stats;
    hipEvent_t take();
    void begin(hipStream_t s, const char *name, double bytes, double onchip = 0.0);
    void end(hipStream_t s);
    void resolve();
    void reset();
    ~Profile();
};

// ---- sparse storage -------------------------------------------------------------------
// One wave works one item: a run of at most ITEM_NNZ nonzeros of one outer vector.
// Outer vectors longer than that are cut into several items whose partial results go
// through a slab and are summed in item order (deterministic, no float atomics).
constexpr uint32_t ITEM_NNZ = 8192;
constexpr uint32_t NO_SLAB = 0xFFFFFFFFu;

struct Item {
    uint64_t start; // offset of the first nonzero in indices/values
    uint32_t row;   // outer vector id
    uint32_t len;   // number of nonzeros
    uint32_t slab;  // slab row for the partial result, NO_SLAB when the item is the whole vector
    uint32_t _pad;
};

struct MultiRow {
    uint32_t row;
    uint32_t first_slab;
    uint32_t count;
    uint32_t _pad;
};

// tiles.hip: the split of an orientation (under one map) into fixed record positions per (outer vector, panel tile) and an
// overflow matrix — what the hybrid LDS-tile + gather product walks
struct TileLayout;
void tile_layout_free(TileLayout *t);
void tile_layout_forget_weights(TileLayout *t);
void tile_layout_stats(const TileLayout *t, uint64_t out[3]); // positions per pass, served nonzeros, overflow nonzeros

// A compressed orientation: n_outer vectors over n_inner positions.
struct SparseCopy {
    uint64_t n_outer = 0, n_inner = 0, nnz = 0;
    DevBuf<uint64_t> indptr;
    DevBuf<uint32_t> indices, values;
    DevBuf<Item> items;
    DevBuf<MultiRow> multi;
    uint32_t n_items = 0, n_multi = 0, n_slab = 0;
    uint32_t max_value = 0;  // largest count (0: not computed yet; counts are >= 1) — bound of the mapped values for col_moments_kernel
    DevBuf<uint32_t> bounds; // L2-blocked gather: offset of the first nonzero >= b*1024 within each outer vector
    // L2-blocked gather, launch order: outer vectors by descending length (longest first) and the sorted lengths
    // on the host (how many vectors are "hot" for a given step count is a binary search)
    DevBuf<uint32_t> order;
    std::vector<uint32_t> sorted_len;
    std::shared_ptr<TileLayout> tiles; // built on first use of the hybrid tile product under a given map (tiles.hip)
    mutable bool tile_flow_refused = false;    // the flow layout cannot serve this copy (a map that does not separate, more large counts than the one-walk build's list holds): the dense layout takes over
    uint64_t tile_rejected_shape = 0;  // the tile shape (and overflow limit) whose layout was not worth building for this copy (0: none)
    int tsig_n = -1;                   // ... the map the last eligible product came with (auto path: build on the second sighting)
    uint32_t tsig_id[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int tsig_outer[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // Materialized map values (kernels.hip, "materialized prefix"): f64 per nonzero, in this copy's order, of the first
    // `fsig_n` links of a map chain — kept for the copy with few, long outer vectors, where evaluating the chain costs a
    // scattered 8-byte gather per nonzero per product (the per-barcode scale while walking a gene's vector).
    DevBuf<double> fvals;
    int fsig_n = 0;
    uint32_t fsig_id[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // MapOp ids of the materialized links (never reused)
    int fsig_outer[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ... and how each indexes its arrays in this copy
    void build_items(hipStream_t s);
};

struct ShardInfo {
    uint32_t rank = 0, world = 1;
    uint64_t outer_begin = 0, outer_global = 0;
    scanrs_allreduce_fn allreduce = nullptr; // host-supplied hook (tests; a caller with its own transport)
    void *ctx = nullptr;
    scanrs_comm *comm = nullptr; // the library's own transport (RCCL or the single-process group); not owned
    // a transport given at world == 1 is still served (single-rank RCCL runs exercise the exchange steps)
    bool active() const { return world > 1 || allreduce != nullptr || comm != nullptr; }
};

// Storage shared by a handle and its views (AdaptiveMat::view / t share `&[AdaptiveVec]`).
// Pinned host staging buffers, kept across handles (capi.cpp): take the smallest idle one of at least `bytes` or pin a new one.
void *pinned_take(size_t bytes, size_t *got);
void pinned_give(void *p, size_t bytes) noexcept;

#ifndef SCANRS_TILE_FLOW_DEFAULT
#define SCANRS_TILE_FLOW_DEFAULT 0
#endif
struct Storage {
    uint64_t rows = 0, cols = 0;
    int storage = SCANRS_CSR; // orientation of `primary`: CSR -> outer = rows
    SparseCopy primary;
    SparseCopy other; // the transposed orientation, built on first use
    std::atomic<bool> has_other{false}; // written by the helper thread too (SideBuild); readers that must not depend on its timing use other_settled
    bool other_settled = false;         // calling thread only: `other` is known to exist at a point of the program that does not depend on the helper's speed (copy_with_outer_rows)
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr; // small dense work that overlaps a sparse pass (svd_bk's cross-block orthogonalisation)
    hipStream_t aux();
    hipStream_t aux2_stream = nullptr; // svd_bk's early projection GEMMs (bulk work nothing waits for until the end of the iterations)
    hipStream_t aux2();
    // scratch key of the stream work is being queued on: the auxiliary stream runs beside the main one and must not share its temporaries
    std::string skey(const char *k) const {
        if (aux_stream && stream == aux_stream) return std::string(k) + "@aux";
        if (aux2_stream && stream == aux2_stream) return std::string(k) + "@aux2";
        return std::string(k);
    }
    hipStream_t ov_stream = nullptr; // the gather over the overflow part of a tile layout runs here, beside the tile kernel (tiles.hip)
    hipEvent_t ev_in = nullptr, ev_ov = nullptr;
    hipStream_t ov();
    // Side build (round 4): the copy and the tile layout that the SECOND product of a solver needs are made by a helper thread on a
    // stream of its own while the main thread prepares and runs the first product (capi.cpp, prepare_second_orientation)
    struct SideBuild;
    SideBuild *side = nullptr;
    void side_join_if(const SparseCopy *target, bool need_layout); // waits until the helper is done with `target` (its copy, or also its layout; nullptr: with everything); rethrows its failure
    uint64_t t_side_wait_us = 0, t_layout_us = 0, t_start_panel_us = 0, t_delivery_us = 0; // first-call accounting (scanrs_mat_get_counter)
    Scratch scratch;
    // GF(2) jump tables of the device-side seeded-panel generator (solver.cpp / omega_jump_kernel)
    DevBuf<uint64_t> jump_tab;
    uint64_t jump_d = 0;
    int jump_npow = 0;
    // grow-only pinned host staging buffer (seeded start panels): no page faults, DMA-speed uploads
    void *host_stage = nullptr;
    size_t host_stage_bytes = 0;
    // (from / back to a small process-wide pool, capi.cpp: pinning 64 MB takes several ms, and scanrs_init can do it ahead of the first call)
    void *pinned(size_t bytes) {
        if (bytes > host_stage_bytes) {
            if (host_stage) pinned_give(host_stage, host_stage_bytes);
            host_stage = nullptr;
            host_stage_bytes = 0;
            host_stage = pinned_take(bytes, &host_stage_bytes);
        }
        return host_stage;
    }
    Profile prof;
    ShardInfo shard; // sharding of primary's outer dimension
    // PcaResult of the last svd_bk / svd_rand call, as it lies in the solver's scratch (scanrs_pca_result_device)
    struct PcaDev {
        const double *u = nullptr, *v = nullptr;
        uint32_t ld_u = 0, ld_v = 0, k = 0;
        uint64_t rows_u = 0, rows_v = 0;
    } pca_dev;
    int spmm_path = 0;                    // 0 auto, 1 plain gather, 2 L2-blocked gather, 3 hybrid: LDS-staged tiles + gather of the overflow (tiles.hip)
    uint32_t tile_k = 2, tile_s = 32;     // hybrid product: record positions per (outer vector, visit), outer vectors per wave
    uint32_t tile_ku = 1;                 // ... unit positions among them (0 or 1; used with tile_k 2): count-1 nonzeros, added without a weight
    uint32_t tile_t = 48, tile_b = 4;     // ... panel rows per tile (<= 24 tile_k) and tile buffers in the LDS ring (tile_t * tile_b <= 192)
    int side_build = 1;                   // solvers: the second product's copy / tile layout on a helper thread beside the first pass (0: built on demand by the caller's thread)
    int tile_split = 1;                   // tile layout: slots per outer vector from its density (several for dense vectors, none — all overflow — for very sparse ones); 0: one slot per vector
    double tile_split_x = 1.8;            // ... nonzeros per panel tile a slot is sized for
    double tile_split_min = 0.5;          // ... vectors below this many nonzeros per tile get no slot
    int tile_build_one_pass = 1;          // wave-level layout builder: records and overflow in ONE walk over the matrix + a compaction (0: counting pass, then fill pass)
    int tile_weights_wide = 1;            // weight refresh of a unit-mode layout: four positions per thread with wide loads / stores (0: one position per thread)
    int tile_dense = 1;                   // tile layout of the default shape: 1 = dense record streams, accumulators picked through VGPR index mode (round 5, tiles_dense.inc); 0 = fixed positions per (slot, visit) (round 4)
    int tile_sort_slots = 2;              // dense tile layout: 2 = the slots sorted by load and DEALT over the groups (every group one slot of every load stratum: equal items, lockstep through the panel's tiles), 1 = consecutive ranks per group, 0 = vector order
    int tile_emit_staged = 1;             // dense tile layout: the stream emission keeps its tables in LDS (0: searches them in global memory, the form for parts of > 4 000 tiles)
    int tile_flow = SCANRS_TILE_FLOW_DEFAULT;                    // dense tile layout with the map evaluated in the kernel: 1 = the FLOW form (tiles_flow.inc): tiles of 32 rows in a ring of 6, one record stream per wave cut into rounds of 64 positions whatever tiles they belong to, the ring handed over at ticks through counters in LDS; 0 = the round-5 form (one barrier per tile of 48 rows, a round per visit)
    int tile_wtab = 1;                    // dense tile layout, folded separable map: the product kernel gathers a position's weight from the map's table by the record itself - the map evaluated inside the kernel, no weight stream (0: one f64 per position, refreshed per normalize)
    int tile_fold = 1;                    // dense tile layout, separable map: the factor of the side without the nonlinear links stays out of the per-position weights (0: both factors in every weight)
    uint64_t tile_big_list_cap = 0;       // dense tile layout, one-walk build: capacity of the list of nonzeros with counts above 255 (0: max(4 M, nnz / 64)); beyond it the two-walk build takes over
    int tile_one_walk = 1;                // dense tile layout: built in one walk over the matrix (0: count walk + fill walk; the same layout)
    int tile_builder = 1;                 // layout builder: 1 = wave-level (a lane per vector, visits in lock-step, rows written whole; default shape only), 0 = per-thread walk
    uint32_t tile_build_waves = 0;        // ... its waves per CU through a dummy LDS allocation (0: no cap — measured the same at 8, 16, 32 and without; and a builder that asks for LDS cannot run beside the persistent tile kernel of a first pass)
    size_t ov_tile_bytes = 0;             // hybrid product: panel slice per step of the overflow gather (0 = twice l2_tile_bytes)
    double tile_max_overflow = 0.35;      // auto path: an orientation whose layout would leave more than this share of the nonzeros to the overflow gather stays on the gather kernels
    int tile_auto = 1;                    // auto path may use the hybrid product (0: only spmm_path 3 does)
    const double *tile_xc_src = nullptr;  // the panel whose compact copy "tile_xc" already holds (set and cleared by mat_apply around one product)
    uint32_t tile_xc_l = 0;
    int tile_hint = 0;                    // > 0 while a solver that repeats the same products is running (svd_bk, svd_rand)
    int tile_overlap = 1;                 // hybrid product: 1 = the overflow gather runs beside the tile kernel (own stream); 0 = after it (measurement)
    int panel_precision = 0;              // 0: f64 panels (default); 1: gathered panels rounded to f32, f64 sums (opt-in)
    size_t l2_tile_bytes = 3584u << 10;   // panel slice per step of the L2-blocked gather (4 MB L2 per XCD): whole 1024-row base tiles up to 3.5 MB — 4 tiles (3.2 MB) at 100 columns, 3 (2.9 MB) at 122; measured 40.55 / 39.71 ms per pass against 41.30 / 40.23 with 3 tiles and 40.86 / 39.61 with 5, and 4 tiles of 122 columns (3.9 MB) lose 1.8 ms
    int spmm_order = 1;                   // L2-blocked gather: launch outer vectors longest first: 0 never, 1 auto, 2 always (SCANRS_SPMM_ORDER)
    uint32_t hot_segment = 512;           // ... and give a workgroup to vectors with >= this many nonzeros per step (0 = never)
    uint64_t blocked_min_nnz = 1ull << 22; // auto: matrices below this stay on the plain gather kernel
    int slice_walk = 1;                   // Ix1 products / moments on the short-outer copy stage the inner-indexed arrays in LDS slices
    int spmv_lds = 1;                     // Ix1 products on the long-outer copy stage the vector in LDS parts
    int dense_side_no_lds = 0;            // dense kernels queued on the side streams use the register-only MFMA forms (they can run BESIDE the persistent tile kernel, which holds the LDS)
    int overlap = 1;                      // small dense work of the solvers on a second stream beside the sparse passes
    const int *skip_flag = nullptr;       // device flag the dense kernels launched now test first (nonzero: return at once) — set around the queued passes of a device-side orthonormalisation
    uint64_t orth_fallbacks = 0;          // orthonormalisations that ended in the host Gram-Schmidt for rank-deficient panels (solver.cpp)
    uint64_t bk_host_retries = 0;         // svd_bk calls that fell back from the device-side factorizations to the host path (scanrs_mat_get_counter)
    int col_moments = 1;                  // per-axis sums of a log-normalized map from the copy whose OUTER vectors are summed over (per-cell table + LDS fixed-point scatter) when eligible; 0: always the ordinary pass
    int device_factor = 1;                // svd_bk: CholeskyQR factors and the coefficient bookkeeping on the device, no host round trip per orthonormalisation (0: host)
    unsigned d2h_threads = 4;             // host threads that empty the pinned ring of a large result download
    int tile_spare_cus = 1;               // the persistent tile kernel launches only as many workgroups as its number of item rounds needs; the CUs left over serve the side streams during the pass (0: a workgroup on every CU)
    int spmv_row_table = 1;               // Ix1 products over many short outer vectors: a map of the count and the outer position alone is looked up from a per-vector table instead of materialized per nonzero (0: materialized values)
    int gemm_direct = 1;                  // dense X W: operands straight from memory into the MFMA registers (0: the LDS-tiled kernels)
    double reuse_cmax = 1e5;              // svd_bk: coefficient bound above which a projection column is recomputed directly
    int materialize = 1;                  // keep the map prefix's values per nonzero on the short-outer copy (SCANRS_MATERIALIZE=0: off)
    ~Storage();
    // the copy whose outer dimension is the base matrix's rows (true) or cols (false)
    SparseCopy &copy_with_outer_rows(bool outer_rows);
};

// ---- lazy map (sqz::MatrixMap chain) ------------------------------------------------------
enum { OP_INTO = 0, OP_SCALE_AXIS = 1, OP_LN_1P = 2, OP_LOG2_1P = 3, OP_LOG10_1P = 4, OP_SQUARE = 5, OP_BINOM_DEV = 6, OP_BINOM_PEARSON = 7 };
constexpr int MAX_OPS = 8; // SparseCopy::fsig_* are sized to this

uint32_t next_map_op_id(); // capi.cpp: never reused (1 ..), identifies a link and the arrays it holds
struct MapOp {
    int kind = OP_INTO;
    int axis = 0;
    bool swap = false; // under an odd number of TransposeMap wrappers
    std::shared_ptr<DevBuf<double>> a, b;
    uint32_t id = next_map_op_id(); // copies of a link (views) keep the id: same arrays, same values
};

// what the kernels see (by value)
struct DevOp {
    int kind;
    int a_outer; // index `a` by the outer (1) or inner (0) position of the copy being walked
    int b_outer;
    uint32_t id; // MapOp::id
    const double *a;
    const double *b;
};
struct DevMap {
    int n;
    int _pad;
    DevOp ops[MAX_OPS];
};

inline uint32_t even_up(uint32_t x) { return (x + 1u) & ~1u; }

} // namespace scanrs

// The opaque handle of the C ABI.
struct scanrs_mat {
    std::shared_ptr<scanrs::Storage> st;
    bool transposed = false; // this view is the base matrix transposed
    std::vector<scanrs::MapOp> ops;
    uint32_t off_rank = 0;
    std::shared_ptr<scanrs::DevBuf<double>> off_u; // rows x rank (view coordinates)
    std::shared_ptr<scanrs::DevBuf<double>> off_v; // rank x cols
    double target_umi = 0.0;

    uint64_t rows() const { return transposed ? st->cols : st->rows; }
    uint64_t cols() const { return transposed ? st->rows : st->cols; }
    // is the (view) outer dimension that contracts in V*x sharded?  (see Storage::shard)
    scanrs::DevMap dev_map(bool outer_is_view_row) const;
};

namespace scanrs {

// ---- kernels.hip launchers ----------------------------------------------------------------
// out[n_outer x l] = S * X (+ a * w) where S is `cp` seen through `map`.
// off_a: n_outer x rank (row-major) or null; off_w: rank x l (ld = ldw).
void launch_spmm_f64(Storage &st, SparseCopy &cp, const DevMap &map, const double *X, uint32_t ldx, uint32_t l,
                     double *out, uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w, uint32_t ldw);
void launch_spmm_u32(Storage &st, const SparseCopy &cp, const uint32_t *X, uint32_t ldx, uint32_t l, uint32_t *out,
                     uint32_t ldo);
// tiles.hip
bool spmm_tiles_ok(const Storage &st, const SparseCopy &cp, uint32_t ldx, uint32_t l);
bool tile_shape_ok(uint32_t K, uint32_t S, uint32_t T, uint32_t B);
bool spmm_tiles_auto(Storage &st, SparseCopy &cp, const DevMap &map);
// the structure of the layout (no weights) when the auto path would take it for this copy: built on stream s; false when the copy
// is not eligible or its layout is not worth having (then remembered in cp.tile_rejected_shape)
bool tile_layout_build_auto(Storage &st, SparseCopy &cp, hipStream_t s);
void launch_spmm_tiles(Storage &st, SparseCopy &cp, const DevMap &map, const double *X, uint32_t ldx, uint32_t l, double *out,
                       uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w, uint32_t ldw);
// per-outer-vector reductions. mode 0: sum of raw u32 counts; 1: sum of mapped values; 2: sum and sum of squares.
void launch_row_reduce(Storage &st, SparseCopy &cp, const DevMap &map, int mode, uint32_t *out_u32, double *out_sum,
                       double *out_sumsq);
// w[rank x l] (ld ldw) = B^T X, B: n x rank row-major, X: n x l (ld ldx)
void launch_weighted_colsum(Storage &st, const double *B, uint32_t rank, const double *X, uint32_t ldx, uint64_t n,
                            uint32_t l, double *w, uint32_t ldw, double *Xc = nullptr, uint32_t ldc = 0);
// where the dense tile product of (cp, l) will stage its compact panel copy from — when the caller can fill it while it reads the
// panel anyway (mat_apply: the column sums of the offset term) — or nullptr (another path, a layout about to be rebuilt, odd l)
double *tile_panel_copy_target(Storage &st, SparseCopy &cp, uint32_t l);
// C[n x m] (row-major, ld = m) = X^T Y over `rows` rows (f64 MFMA, slab + ordered reduce)
void launch_gram(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *Y, uint32_t ldy, uint32_t m,
                 uint64_t rows, double *C);
// Out[rows x m] = beta * Cin + alpha * X[rows x n] * W[n x m]   (W device, row-major ld = ldw). Cin may equal Out.
void launch_gemm_nn(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *W, uint32_t ldw, uint32_t m,
                    uint64_t rows, double alpha, double beta, const double *Cin, uint32_t ldc, double *Out, uint32_t ldo);
// dense.hip: LDS-tiled MFMA versions for big panels
bool gram_tiled_ok(uint32_t n, uint32_t m, uint64_t rows);
bool gemm_tiled_ok(uint32_t n, uint32_t m, uint64_t rows);
void launch_gram_tiled(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *Y, uint32_t ldy, uint32_t m,
                       uint64_t rows, double *C);
bool gemm_direct_ok(const double *X, uint32_t ldx, uint32_t n, uint32_t m, uint64_t rows);
void launch_gemm_direct(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *W, uint32_t ldw, uint32_t m, uint64_t rows,
                        double alpha, double beta, const double *Cin, uint32_t ldc, double *Out, uint32_t ldo);
void launch_gemm_tiled(Storage &st, const double *X, uint32_t ldx, uint32_t n, const double *W, uint32_t ldw, uint32_t m,
                       uint64_t rows, double alpha, double beta, const double *Cin, uint32_t ldc, double *Out, uint32_t ldo);
void launch_col_scale_dev(Storage &st, double *dst, uint32_t ldd, const double *src, uint32_t lds, uint64_t rows, const double *nsq);
void launch_col_axpy_dev(Storage &st, double *y, uint32_t ldy, const double *x, uint32_t ldx, uint64_t rows, const double *nsq);
void launch_col_scale(Storage &st, double *dst, uint32_t ldd, const double *src, uint32_t lds, uint64_t rows, double alpha);
void launch_col_axpy(Storage &st, double *y, uint32_t ldy, const double *x, uint32_t ldx, uint64_t rows, double alpha);
void launch_copy_cols(Storage &st, const double *src, uint32_t lds, double *dst, uint32_t ldd, uint64_t rows, uint32_t l);
void launch_fill_f64(Storage &st, double *p, uint64_t n, double v);
void launch_fill_hash(Storage &st, double *p, uint32_t ld, uint64_t rows, uint64_t row0, uint32_t c0, uint32_t nc, uint64_t seed);
void launch_omega_jump(Storage &st, const uint64_t *d_jpow, int n_pow, const uint64_t s[4], uint64_t d, uint64_t total, double *out,
                       uint32_t ld, uint64_t seq_cols, bool transpose);
void launch_transpose(Storage &st, const double *src, uint64_t rows, uint64_t cols, double *dst, uint32_t ldd);
void launch_permute_cols(Storage &st, const double *src, uint32_t lds, double *dst, uint32_t ldd, uint64_t rows,
                         const uint32_t *d_idx, uint32_t n_idx, bool scatter);
void launch_finish_moments(Storage &st, const double *sum, const double *sumsq, uint64_t n, double m, int given_scale,
                           const double *scale_in, double *mean_over_scale_neg, double *inv_scale, double *scale_out);
void launch_u32_to_scale(Storage &st, const uint32_t *counts, uint64_t n, double target, double *out);
void launch_hist12(Storage &st, const uint32_t *v, uint64_t n, uint32_t shift, uint32_t digit_mask, uint32_t prefix_mask,
                   uint32_t prefix, unsigned long long *hist);
void launch_sum_f64(Storage &st, const double *x, uint64_t n, double *out);
void validate_copy(Storage &st, SparseCopy &cp, uint64_t *zeros, uint64_t *bad);
void launch_densify(Storage &st, const SparseCopy &cp, const DevMap &map, bool outer_is_view_row, uint64_t cols_v,
                    double *out);
void launch_binom_uv(Storage &st, int kind, const double *n, uint64_t ncols, const double *rowsum, uint64_t nrows,
                     double total, double *pi, double *u, double *v);
// build `dst` = transpose of `src` (stable: inner vectors keep ascending outer order)
void build_transposed_copy(Storage &st, const SparseCopy &src, SparseCopy &dst, hipStream_t stream = nullptr);
// drop stored zeros, compact (device). Returns new nnz.
void compact_nonzeros(Storage &st, SparseCopy &cp);

// ---- host_linalg.cpp --------------------------------------------------------------------------
// Upper Cholesky G = R^T R of an n x n SPD matrix (row-major, in place: upper triangle = R). false if not SPD.
bool launch_col_moments(Storage &st, SparseCopy &cp, const DevMap &map, int mode, double *out_sum, double *out_sumsq);
bool chol_rinv_ok(uint32_t n);
void launch_chol_rinv(Storage &st, const double *G, uint32_t n, uint64_t rows, int pass, bool check_only, int *ctl, double *Rinv, double *info);
void launch_absmax_flag(Storage &st, const double *C, uint32_t count, double limit, int *ctl, double *info);
bool chol_upper(double *g, int n);
// in place inverse of an upper-triangular matrix
void inv_upper(double *r, int n);
// symmetric eigen-decomposition: on return w[0..n) descending, z row-major with z[i*n + j] = component i of vector j.
bool sym_eig(const double *a, int n, double *w, double *z);
// k leading eigenpairs only: w[0..k) descending, z row-major n x k
bool sym_eig_topk(const double *a, int n, int k, double *w, double *z);

// ---- solver.cpp ------------------------------------------------------------------------------------
int pca_bk(scanrs_mat *m, uint32_t k, double k_multiplier, uint32_t n_iter, uint64_t seed, const double *omega,
           const scanrs_snoop *snoop, double *u, double *s, double *v);
int pca_rand(scanrs_mat *m, uint32_t k, double l_multiplier, uint32_t n_iter, uint64_t seed, const double *omega, double *u,
             double *s, double *v);
int pca_irlba(scanrs_mat *m, uint32_t nu, double tol, uint32_t max_iter, const double *v0, const scanrs_snoop *snoop,
              double *u, double *s, double *v, uint32_t *mprod);
void omega_fill(uint64_t seed, uint64_t count, double *out);
void jump_tables_prefetch(); // starts the background computation of the start panel generator's jump tables (once per process)

// operator-level helpers shared by capi.cpp and solver.cpp
// out[rows_v x l] = V * X  (transpose: out[cols_v x l] = V^T * X), offsets and shard reduction included.
void mat_apply(scanrs_mat *m, bool transpose, const double *dX, uint32_t ldx, uint32_t l, double *dOut, uint32_t ldo);
bool mat_tiles_ready(scanrs_mat *m, bool transpose);
void prepare_second_orientation(scanrs_mat *m, bool transpose_second, bool solver_follows = false);
void sort_outer_vectors(Storage &st, SparseCopy &cp);
void launch_scale_rows(Storage &st, const double *X, uint32_t ldx, uint64_t rows, uint32_t l, const double *a, double *Xs);
// knn.hip
void knn_host(const double *queries, uint64_t n_q, const double *points, uint64_t n_p, uint32_t d, uint32_t k, bool skip_same_index,
              uint32_t *out);
// the same with both point sets already in device memory (row-major, leading dimensions in elements); `out` is a host array
void knn_device(const double *d_queries, uint32_t ld_q, uint64_t n_q, const double *d_points, uint32_t ld_p, uint64_t n_p, uint32_t d,
                uint32_t k, bool skip_same_index, uint32_t *out);
// decode.hip
uint64_t decode_adaptive_vectors(const scanrs_adaptive_vec *vecs, uint64_t n_vecs, uint64_t vec_len, DevBuf<uint64_t> &indptr,
                                 DevBuf<uint32_t> &indices, DevBuf<uint32_t> &values);
// comm.cpp
struct LocalGroup;
void comm_allreduce(Storage &st, scanrs_comm *c, void *d, uint64_t count, int dtype); // dtype 0 = f64, 1 = u64; on st.stream
void comm_abort(scanrs_comm *c);
std::shared_ptr<LocalGroup> local_group_make(uint32_t world);
void local_group_reset(LocalGroup &g);
scanrs_comm *comm_make_local(const std::shared_ptr<LocalGroup> &g, uint32_t rank);
void allreduce_f64(Storage &st, double *d, uint64_t count);
void allreduce_u64(Storage &st, unsigned long long *d, uint64_t count);
// is the view-row dimension the sharded one?
bool rows_sharded(const scanrs_mat *m);
bool cols_sharded(const scanrs_mat *m);

} // namespace scanrs
