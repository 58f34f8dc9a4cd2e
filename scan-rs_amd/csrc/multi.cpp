// multi.cpp — the single-process multi-GPU form of the boundary (`mat_create(..., n_gpus)`).
// Cell Ranger is one process (tools/src/bin/cmd.rs:61-70): it hands over the whole matrix once and calls normalize
// and run_pca once. Here the outer vectors are range-partitioned by nonzeros over `n_shards` devices, every shard is an
// ordinary scanrs_mat handle on its own device driven by its own host thread for the duration of a call, and the
// exchange steps go through the library's single-process group (comm.cpp). Several shards may share one device
// (`devices` repeats an id): that is how a 1-GPU box exercises the whole path.
#include <functional>
#include <thread>

#include "common.hpp"

struct scanrs_multi {
    uint64_t rows = 0, cols = 0;
    int storage = SCANRS_CSR;
    std::vector<int> devices;
    std::vector<uint64_t> bounds; // n_shards + 1, in outer vectors
    std::vector<scanrs_mat *> shards;
    std::vector<scanrs_comm *> comms;
    std::shared_ptr<scanrs::LocalGroup> group;
};

using namespace scanrs;

namespace {

struct ShardStatus {
    int code = SCANRS_OK;
    std::string msg;
};

// run f(i) for every shard on its own thread with its device current; first failure wins (cancellation over the
// secondary "another shard failed")
template <typename F>
int fan_out(scanrs_multi *mm, F &&f) {
    const size_t n = mm->shards.size();
    std::vector<ShardStatus> st(n);
    std::vector<std::thread> th;
    // a failed or cancelled earlier operation aborted its own barriers only: the handle stays usable (no shard thread runs here)
    if (mm->group) local_group_reset(*mm->group);
    for (size_t i = 0; i < n; i++) {
        th.emplace_back([&, i] {
            if (hipSetDevice(mm->devices[i]) != hipSuccess) {
                st[i].code = SCANRS_ERR_DEVICE;
                st[i].msg = "hipSetDevice failed";
            } else {
                st[i].code = f(i);
                if (st[i].code != SCANRS_OK) st[i].msg = scanrs_last_error();
            }
            if (st[i].code != SCANRS_OK && i < mm->comms.size()) comm_abort(mm->comms[i]); // wake the others out of their barriers
        });
    }
    for (auto &t : th) t.join();
    int first = SCANRS_OK;
    size_t who = 0;
    for (size_t i = 0; i < n; i++) {
        if (st[i].code == SCANRS_OK) continue;
        const bool secondary = st[i].msg.find("another shard") != std::string::npos;
        if (first == SCANRS_OK || (!secondary && st[who].msg.find("another shard") != std::string::npos)) {
            first = st[i].code;
            who = i;
        }
    }
    if (first != SCANRS_OK) set_error("shard %zu: %s", who, st[who].msg.c_str());
    return first;
}

} // namespace

extern "C" {

int scanrs_multi_create(uint64_t rows, uint64_t cols, int storage, const uint64_t *indptr, const uint32_t *indices,
                        const uint32_t *values, uint32_t n_shards, const int *devices, scanrs_multi **out) {
    try {
        if (!out) fail(SCANRS_ERR_ARGUMENT, "null output handle");
        *out = nullptr;
        if (!indptr || n_shards == 0 || n_shards > 16) fail(SCANRS_ERR_ARGUMENT, "1 <= n_shards <= 16 and a triplet are required");
        if (storage != SCANRS_CSR && storage != SCANRS_CSC) fail(SCANRS_ERR_ARGUMENT, "storage must be 0 (CSR) or 1 (CSC)");
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) fail(SCANRS_ERR_DEVICE, "no gfx950 (MI355X) device is usable from this process");
        auto mm = std::make_unique<scanrs_multi>();
        mm->rows = rows;
        mm->cols = cols;
        mm->storage = storage;
        for (uint32_t i = 0; i < n_shards; i++) {
            const int d = devices ? devices[i] : (int)i;
            if (d < 0 || d >= n_dev) fail(SCANRS_ERR_ARGUMENT, "device %d of shard %u does not exist (%d visible)", d, i, n_dev);
            mm->devices.push_back(d);
        }
        const uint64_t n_outer = storage == SCANRS_CSR ? rows : cols;
        mm->bounds.resize(n_shards + 1);
        if (scanrs_plan_shards(indptr, n_outer, n_shards, mm->bounds.data()) != SCANRS_OK) throw Failure{SCANRS_ERR_ARGUMENT};
        // peer access between every pair of distinct devices (the one-shot all-reduce reads peers' partial sums in place)
        for (uint32_t i = 0; i < n_shards; i++)
            for (uint32_t j = 0; j < n_shards; j++) {
                if (mm->devices[i] == mm->devices[j]) continue;
                int can = 0;
                SCANRS_HIP(hipDeviceCanAccessPeer(&can, mm->devices[i], mm->devices[j]));
                if (!can) fail(SCANRS_ERR_DEVICE, "devices %d and %d cannot map each other's memory", mm->devices[i], mm->devices[j]);
                SCANRS_HIP(hipSetDevice(mm->devices[i]));
                const hipError_t e = hipDeviceEnablePeerAccess(mm->devices[j], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) SCANRS_HIP(e);
                (void)hipGetLastError();
            }
        mm->group = local_group_make(n_shards);
        mm->shards.assign(n_shards, nullptr);
        for (uint32_t i = 0; i < n_shards; i++) mm->comms.push_back(comm_make_local(mm->group, i));
        scanrs_multi *raw = mm.get();
        const int rc = fan_out(raw, [&](size_t i) {
            const uint64_t lo = raw->bounds[i], hi = raw->bounds[i + 1];
            std::vector<uint64_t> ip(hi - lo + 1);
            for (uint64_t o = lo; o <= hi; o++) ip[o - lo] = indptr[o] - indptr[lo];
            const uint64_t r = storage == SCANRS_CSR ? hi - lo : rows, c = storage == SCANRS_CSR ? cols : hi - lo;
            int rc2 = scanrs_mat_create(r, c, storage, ip.data(), indices + indptr[lo], values + indptr[lo], &raw->shards[i]);
            if (rc2 != SCANRS_OK) return rc2;
            return scanrs_mat_set_shard_comm(raw->shards[i], raw->comms[i], (uint32_t)i, (uint32_t)raw->shards.size(), lo, n_outer);
        });
        if (rc != SCANRS_OK) {
            scanrs_multi_free(mm.release());
            return rc;
        }
        *out = mm.release();
        return SCANRS_OK;
    } catch (const Failure &e) {
        return e.code;
    } catch (const std::exception &e) {
        set_error("internal error: %s", e.what());
        return SCANRS_ERR_DEVICE;
    }
}

void scanrs_multi_free(scanrs_multi *mm) {
    if (!mm) return;
    for (size_t i = 0; i < mm->shards.size(); i++) {
        (void)hipSetDevice(mm->devices[i]);
        scanrs_mat_free(mm->shards[i]);
    }
    for (auto *c : mm->comms) scanrs_comm_free(c);
    delete mm;
}

int scanrs_multi_n_shards(const scanrs_multi *mm, uint32_t *n) {
    if (!mm || !n) return SCANRS_ERR_ARGUMENT;
    *n = (uint32_t)mm->shards.size();
    return SCANRS_OK;
}

int scanrs_multi_shard(scanrs_multi *mm, uint32_t i, scanrs_mat **shard, int *device, uint64_t *outer_begin, uint64_t *outer_end) {
    if (!mm || i >= mm->shards.size()) return SCANRS_ERR_ARGUMENT;
    if (shard) *shard = mm->shards[i];
    if (device) *device = mm->devices[i];
    if (outer_begin) *outer_begin = mm->bounds[i];
    if (outer_end) *outer_end = mm->bounds[i + 1];
    return SCANRS_OK;
}

int scanrs_multi_normalize(scanrs_multi *mm, int normalization, const uint32_t *size_factors) {
    if (!mm) return SCANRS_ERR_ARGUMENT;
    return fan_out(mm, [&](size_t i) {
        // size factors are per column: the local slice when the columns are the sharded dimension
        const uint32_t *sf = size_factors;
        if (sf && mm->storage == SCANRS_CSC) sf += mm->bounds[i];
        return scanrs_normalize(mm->shards[i], normalization, sf);
    });
}

// u: rows x k, s: k, v: cols x k (row-major, caller-allocated; u and/or v may be null). The factor on the sharded side
// is assembled from the shards' rows, the replicated one is taken from shard 0.
static int multi_pca(scanrs_multi *mm, uint32_t k, double *u, double *s, double *v, const std::function<int(size_t, double *, double *, double *)> &call) {
    if (!mm || !s) return SCANRS_ERR_ARGUMENT;
    const bool cols_sharded = mm->storage == SCANRS_CSC;
    std::vector<std::vector<double>> s_each(mm->shards.size(), std::vector<double>(k));
    std::vector<double> dummy_rep; // replicated factor of shards > 0 is not downloaded
    return fan_out(mm, [&](size_t i) {
        double *ui = nullptr, *vi = nullptr;
        if (cols_sharded) {
            ui = (i == 0) ? u : nullptr;
            vi = v ? v + mm->bounds[i] * k : nullptr;
        } else {
            ui = u ? u + mm->bounds[i] * k : nullptr;
            vi = (i == 0) ? v : nullptr;
        }
        return call(i, ui, i == 0 ? s : s_each[i].data(), vi);
    });
}

int scanrs_multi_pca_bk(scanrs_multi *mm, uint32_t k, double k_multiplier, uint32_t n_iter, uint64_t seed, const double *omega,
                        const scanrs_snoop *snoop, double *u, double *s, double *v) {
    return multi_pca(mm, k, u, s, v, [&](size_t i, double *ui, double *si, double *vi) {
        scanrs_snoop sn;
        const scanrs_snoop *psn = nullptr;
        if (snoop) { // every shard polls the cancel flag at the same points; only shard 0 reports progress
            sn = *snoop;
            if (i != 0) sn.progress = nullptr;
            psn = &sn;
        }
        return scanrs_pca_bk(mm->shards[i], k, k_multiplier, n_iter, seed, omega, psn, ui, si, vi);
    });
}

int scanrs_multi_pca_rand(scanrs_multi *mm, uint32_t k, double l_multiplier, uint32_t n_iter, uint64_t seed, const double *omega,
                          double *u, double *s, double *v) {
    return multi_pca(mm, k, u, s, v, [&](size_t i, double *ui, double *si, double *vi) {
        return scanrs_pca_rand(mm->shards[i], k, l_multiplier, n_iter, seed, omega, ui, si, vi);
    });
}

// Irlba on the un-centred handle (irlba.rs:59-215; LowRankOffset has no Ix1 Dot impl). v0: optional start vector over ALL columns.
int scanrs_multi_pca_irlba(scanrs_multi *mm, uint32_t nu, double tol, uint32_t max_iter, const double *v0, const scanrs_snoop *snoop,
                           double *u, double *s, double *v, uint32_t *mprod) {
    if (!u || !v) return SCANRS_ERR_ARGUMENT;
    std::vector<uint32_t> mp(mm ? mm->shards.size() : 0, 0);
    const int rc = multi_pca(mm, nu, u, s, v, [&](size_t i, double *ui, double *si, double *vi) {
        scanrs_snoop sn;
        const scanrs_snoop *psn = nullptr;
        if (snoop) {
            sn = *snoop;
            if (i != 0) sn.progress = nullptr;
            psn = &sn;
        }
        const bool cols_sharded = mm->storage == SCANRS_CSC;
        const uint64_t n_loc = mm->bounds[i + 1] - mm->bounds[i];
        // scanrs_pca_irlba wants both factors: the replicated one of shards > 0 goes to a scratch array
        std::vector<double> spare;
        if (!ui) {
            spare.resize((size_t)(cols_sharded ? mm->rows : n_loc) * nu);
            ui = spare.data();
        }
        if (!vi) {
            spare.resize((size_t)(cols_sharded ? n_loc : mm->cols) * nu);
            vi = spare.data();
        }
        const double *v0i = v0 ? (cols_sharded ? v0 + mm->bounds[i] : v0) : nullptr;
        return scanrs_pca_irlba(mm->shards[i], nu, tol, max_iter, v0i, psn, ui, si, vi, &mp[i]);
    });
    if (mprod && !mp.empty()) *mprod = mp[0];
    return rc;
}

// Raw-count log normalisation without the centre / scale step (the only input irlba.rs takes).
int scanrs_multi_log_normalize(scanrs_multi *mm, double umi_count_sum, int log_fn, const uint32_t *size_factors) {
    if (!mm) return SCANRS_ERR_ARGUMENT;
    return fan_out(mm, [&](size_t i) {
        const uint32_t *sf = size_factors;
        if (sf && mm->storage == SCANRS_CSC) sf += mm->bounds[i];
        return scanrs_log_normalize(mm->shards[i], umi_count_sum, log_fn, sf);
    });
}

} // extern "C"
