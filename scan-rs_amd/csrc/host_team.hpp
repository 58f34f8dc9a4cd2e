// A small persistent team of host threads (the threaded parts of sym_eig_topk, the workers that empty the pinned ring of a large
// result download).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

namespace scanrs {

// Creating threads costs more than the
// whole 6 ms it saves (measured 1-3 ms per std::thread in a container), so they are created once, parked on a condition
// variable between uses, and never destroyed (the object is leaked on purpose: no destructor runs while they wait).
// One user at a time: a second concurrent caller (the shards of a MultiMat run their replicated eigenproblems at the same
// time) simply does not get the team and works alone — the arithmetic does not depend on it.
class HostTeam {
  public:
    static HostTeam *acquire(int n_threads) { // nullptr: busy, unavailable, or a forked child without the threads
        static HostTeam *inst = new HostTeam();
        if (inst->pid_ != getpid()) return nullptr; // the threads did not survive a fork
        if (!inst->busy_.try_lock()) return nullptr;
        if (!inst->ensure(n_threads)) {
            inst->busy_.unlock();
            return nullptr;
        }
        return inst;
    }
    void release() { busy_.unlock(); }
    // runs fn(t) on team threads t = 1 .. n-1 (asynchronously); the caller is thread 0 and must call join() afterwards
    void start(int n, std::function<void(int)> fn) {
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = std::move(fn);
            active_ = n;
            remaining_.store(n - 1, std::memory_order_relaxed);
            epoch_++;
        }
        cv_.notify_all();
    }
    void join() {
        int spins = 0;
        while (remaining_.load(std::memory_order_acquire) != 0) {
            if (++spins < 2000)
                __builtin_ia32_pause();
            else
                std::this_thread::yield();
        }
    }

  private:
    HostTeam() : pid_(getpid()) {}
    bool ensure(int n) {
        try {
            while ((int)threads_.size() < n - 1) {
                const int t = (int)threads_.size() + 1;
                threads_.emplace_back([this, t] { loop(t); });
                threads_.back().detach();
            }
        } catch (...) {
            return false;
        }
        return true;
    }
    void loop(int t) {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(int)> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (t >= active_) continue;
                job = job_;
            }
            job(t);
            remaining_.fetch_sub(1, std::memory_order_release);
        }
    }
    pid_t pid_;
    std::mutex busy_, m_;
    std::condition_variable cv_;
    std::vector<std::thread> threads_;
    std::function<void(int)> job_;
    int active_ = 0;
    uint64_t epoch_ = 0;
    std::atomic<int> remaining_{0};
};


} // namespace scanrs
