// comm.cpp — the library's own exchange step (one sum all-reduce per product that contracts over the
// sharded cells, the gene moments, the q x q Gram matrix, three 4096-bin histograms per median).
//
//  * multi-process form (one process per GPU, the way bench.py is launched): RCCL, loaded with dlopen when the first
//    communicator is made — `ncclAllReduce` is enqueued on the handle's own stream, no host synchronisation and no
//    foreign runtime in the loop. The 128-byte unique id travels by whatever channel the host program has
//    (bench.py: a torch.distributed broadcast on the gloo control plane).
//  * single-process form (scanrs_multi_*, SURVEY.md §8b `mat_create(..., n_gpus)`): one host thread per shard inside the
//    library; the all-reduce is a one-shot reduce-scatter + all-gather over peer-mapped memory (every rank's kernel
//    sums ITS slice of all ranks' buffers in rank order and writes it back to all of them): xGMI is point-to-point, so
//    all links carry traffic at once instead of a ring's one; deterministic; works for shards that share a device.
#include <dlfcn.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>

#include "common.hpp"

namespace scanrs {

// ---- RCCL through dlopen ------------------------------------------------------------------------------------------
namespace {
struct NcclUniqueId {
    char internal[128];
};
struct Rccl {
    void *h = nullptr;
    std::string err; // why loading failed (captured right behind the failing call: dlerror() clears itself when read)
    bool reused = false;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;    // optional (scanrs_comm_info)
    int (*CommUserRank)(void *, int *) = nullptr; // optional
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // An RCCL image the host program has already mapped (torch ships and links its own librccl.so.1) is reused: one
        // RCCL per process. Only when there is none is the system's loaded — locally, so that it cannot interpose anybody.
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (r.h) {
                r.reused = true;
                break;
            }
        }
        if (!r.h) {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.h) break;
                if (const char *e = dlerror()) r.err = e;
            }
        }
        if (!r.h) return;
        struct {
            void **slot;
            const char *name;
        } syms[] = {{(void **)&r.GetUniqueId, "ncclGetUniqueId"}, {(void **)&r.CommInitRank, "ncclCommInitRank"},
                    {(void **)&r.CommDestroy, "ncclCommDestroy"}, {(void **)&r.AllReduce, "ncclAllReduce"},
                    {(void **)&r.GetErrorString, "ncclGetErrorString"}};
        r.CommCount = (int (*)(void *, int *))dlsym(r.h, "ncclCommCount");
        r.CommUserRank = (int (*)(void *, int *))dlsym(r.h, "ncclCommUserRank");
        (void)dlerror();
        for (auto &sy : syms) {
            *sy.slot = dlsym(r.h, sy.name);
            if (!*sy.slot && r.err.empty()) {
                const char *e = dlerror();
                r.err = e ? e : (std::string("missing symbol ") + sy.name);
            }
        }
    });
    if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce)
        fail(SCANRS_ERR_DEVICE, "RCCL (librccl.so.1) could not be loaded: %s", r.err.empty() ? "missing symbols" : r.err.c_str());
    return r;
}
void nccl_check(int rc, const char *what) {
    if (rc != 0) {
        Rccl &r = rccl();
        fail(SCANRS_ERR_DEVICE, "%s failed: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    }
}
constexpr int NCCL_SUM = 0, NCCL_UINT64 = 5, NCCL_FLOAT64 = 8; // rccl.h: ncclRedOp_t / ncclDataType_t
} // namespace

// ---- single-process group: barrier with abort + one-shot all-reduce --------------------------------------------------
struct LocalGroup {
    uint32_t world = 0;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    bool aborted = false;
    std::vector<void *> ptrs;
    explicit LocalGroup(uint32_t w) : world(w), ptrs(w, nullptr) {}
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
    // before a new operation, with no shard thread running: an earlier failure or cancellation aborted THAT operation only
    void reset() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = false;
        arrived = 0;
    }
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) fail(SCANRS_ERR_DEVICE, "another shard of the group failed");
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return;
        }
        // bounded like every device wait (a shard thread that died without aborting must not hold the others for ever)
        const bool woke = cv.wait_for(lk, std::chrono::duration<double>(sync_timeout_s()), [&] { return generation != gen || aborted; });
        if (!woke) {
            aborted = true;
            cv.notify_all();
            fail(SCANRS_ERR_DEVICE, "group barrier timed out after %.1f s (sync_timeout_s): %u of %u shards arrived", sync_timeout_s(), arrived, world);
        }
        if (generation == gen) fail(SCANRS_ERR_DEVICE, "another shard of the group failed");
    }
};

} // namespace scanrs

struct scanrs_comm {
    uint32_t rank = 0, world = 1;
    uint64_t n_allreduce = 0, allreduce_bytes = 0; // what went through this communicator (scanrs_comm_info)
    void *nccl = nullptr;                        // multi-process form
    std::shared_ptr<scanrs::LocalGroup> local;   // single-process form
};

namespace scanrs {

void launch_local_allreduce(hipStream_t s, void *const *bufs, uint32_t world, uint32_t rank, uint64_t count, int dtype);

void comm_allreduce(Storage &st, scanrs_comm *c, void *d, uint64_t count, int dtype) {
    if (count == 0) return;
    c->n_allreduce++;
    c->allreduce_bytes += count * 8u;
    if (c->nccl) {
        Rccl &r = rccl();
        nccl_check(r.AllReduce(d, d, (size_t)count, dtype == 0 ? NCCL_FLOAT64 : NCCL_UINT64, NCCL_SUM, c->nccl, st.stream), "ncclAllReduce");
        return;
    }
    if (c->local) {
        LocalGroup &g = *c->local;
        SCANRS_SYNC(st.stream); // my partial sums are complete
        g.ptrs[c->rank] = d;
        g.barrier(); // everybody's are, and every pointer is published
        launch_local_allreduce(st.stream, g.ptrs.data(), g.world, c->rank, count, dtype);
        SCANRS_SYNC(st.stream);
        g.barrier(); // every slice has been written to every buffer
        return;
    }
    if (c->world > 1) fail(SCANRS_ERR_ARGUMENT, "communicator has no transport");
}

void comm_abort(scanrs_comm *c) {
    if (c && c->local) c->local->abort();
}

scanrs_comm *comm_make_local(const std::shared_ptr<LocalGroup> &g, uint32_t rank) {
    auto *c = new scanrs_comm;
    c->rank = rank;
    c->world = g->world;
    c->local = g;
    return c;
}
std::shared_ptr<LocalGroup> local_group_make(uint32_t world) { return std::make_shared<LocalGroup>(world); }
void local_group_reset(LocalGroup &g) { g.reset(); }

} // namespace scanrs

using namespace scanrs;

extern "C" {

int scanrs_comm_get_unique_id(uint8_t *id) {
    try {
        if (!id) fail(SCANRS_ERR_ARGUMENT, "null argument");
        NcclUniqueId u;
        nccl_check(rccl().GetUniqueId(&u), "ncclGetUniqueId");
        memcpy(id, u.internal, sizeof(u.internal));
        return SCANRS_OK;
    } catch (const Failure &e) {
        return e.code;
    }
}

int scanrs_comm_create(const uint8_t *id, uint32_t rank, uint32_t world, scanrs_comm **out) {
    try {
        if (!out) fail(SCANRS_ERR_ARGUMENT, "null argument");
        *out = nullptr;
        if (world == 0 || rank >= world) fail(SCANRS_ERR_ARGUMENT, "bad rank/world");
        if (!id) fail(SCANRS_ERR_ARGUMENT, "null unique id");
        NcclUniqueId u;
        memcpy(u.internal, id, sizeof(u.internal));
        auto c = std::make_unique<scanrs_comm>();
        c->rank = rank;
        c->world = world;
        nccl_check(rccl().CommInitRank(&c->nccl, (int)world, u, (int)rank), "ncclCommInitRank"); // on the calling thread's current device
        *out = c.release();
        return SCANRS_OK;
    } catch (const Failure &e) {
        return e.code;
    }
}

// What the transport itself says about the group (bench.py prints it: the driver's scaling record can then show that RCCL saw all N
// ranks): ranks and this rank as RCCL counts them (ncclCommCount / ncclCommUserRank; the single-process form: its own group), and
// the sum all-reduces that went through so far.
int scanrs_comm_info(scanrs_comm *c, uint32_t *nranks, uint32_t *rank, uint64_t *n_allreduce, uint64_t *allreduce_bytes) {
    try {
        if (!c) fail(SCANRS_ERR_ARGUMENT, "null communicator");
        uint32_t n = c->world, r_ = c->rank;
        if (c->nccl) {
            Rccl &r = rccl();
            int v = 0;
            if (r.CommCount) {
                nccl_check(r.CommCount(c->nccl, &v), "ncclCommCount");
                n = (uint32_t)v;
            }
            if (r.CommUserRank) {
                nccl_check(r.CommUserRank(c->nccl, &v), "ncclCommUserRank");
                r_ = (uint32_t)v;
            }
        }
        if (nranks) *nranks = n;
        if (rank) *rank = r_;
        if (n_allreduce) *n_allreduce = c->n_allreduce;
        if (allreduce_bytes) *allreduce_bytes = c->allreduce_bytes;
        return SCANRS_OK;
    } catch (const Failure &e) {
        return e.code;
    }
}

// diagnostics (CPU test-suite): one thread enters a barrier of a group of `world` shards alone
int scanrs_debug_barrier_alone(uint32_t world) {
    try {
        LocalGroup g(world < 2 ? 2 : world);
        g.barrier();
        return SCANRS_OK;
    } catch (const Failure &e) {
        return e.code;
    }
}

void scanrs_comm_free(scanrs_comm *c) {
    if (!c) return;
    if (c->nccl) {
        try {
            (void)rccl().CommDestroy(c->nccl);
        } catch (const Failure &) {
        }
    }
    delete c;
}

} // extern "C"
