// parallel.hpp -- the host side's only threading primitive: a blocking parallel_for over [0, n).
// Used by the once-per-scene set-up stages (parse, texture decode, triangle boxes, own-list index); the frame path has no host threads.
//
// The ranges run on a process-wide pool of workers that is grown on demand and never torn down: creating a thread costs 30-60 us and threads of one
// process are created one at a time (they share the address-space lock), so the ~500 short-lived threads of a six-texture scene load -- six decoders,
// each with four parallel stages -- cost more than the work they did.  A caller runs one range itself and, while its others are pending, takes queued
// ranges of ANY caller: nested parallel_ranges (a decoder inside the loader's texture loop) therefore cannot deadlock, and a process that lost its
// workers (fork) still completes everything on the calling thread.
#pragma once
#include <algorithm>
#include <pthread.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace rrt {

inline unsigned host_threads() {
    if (const char* e = std::getenv("RRT_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) return (unsigned)std::min(v, 256); }
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hc ? hc : 1u, 128u));     // (the 1 M-triangle .obj, 253 MB of text: 32 ranges 22 ms, 128 ranges 6 ms on a 256-thread host)
}

class HostPool {
public:
    static HostPool& get() { return *slot(); }
    void submit(std::function<void()> task) {
        bool spawn = false;
        {
            std::lock_guard<std::mutex> g(m_);
            q_.push_back(std::move(task));
            if (q_.size() > idle_ && n_workers_ < cap_) { n_workers_++; spawn = true; }   // (idle workers that have not woken yet are still counted: one each)
        }
        if (spawn) {
            try { std::thread([this] { work(); }).detach(); }
            catch (...) { std::lock_guard<std::mutex> g(m_); n_workers_--; }         // no thread to be had: the submitter runs the task when it waits
        }
        cv_.notify_one();
    }
    bool run_one() {                                                               // a queued range, if there is one (called by waiting submitters)
        std::function<void()> t;
        {
            std::lock_guard<std::mutex> g(m_);
            if (q_.empty()) return false;
            t = std::move(q_.front()); q_.pop_front();
        }
        t();
        return true;
    }

private:
    // Leaked on purpose (workers may outlive static destructors).  A forked child has the pool's memory but none of its threads, and possibly its mutex
    // in a locked state: it starts over with a fresh pool.
    static HostPool*& slot() {
        static HostPool* p = [] { pthread_atfork(nullptr, nullptr, [] { slot() = new HostPool; }); return new HostPool; }();
        return p;
    }
    HostPool() { const unsigned hc = std::thread::hardware_concurrency(); cap_ = std::max(1u, std::min(hc ? hc : 1u, 128u)); }
    void work() {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            while (q_.empty()) { idle_++; cv_.wait(lk); idle_--; }
            std::function<void()> t = std::move(q_.front()); q_.pop_front();
            lk.unlock();
            t();
            lk.lock();
        }
    }
    std::mutex m_; std::condition_variable cv_; std::deque<std::function<void()>> q_;
    unsigned n_workers_ = 0, idle_ = 0, cap_ = 1;
};

// f(begin, end, part) on `parts` contiguous ranges of [0, n); parts <= host_threads().  An exception of the EARLIEST range that threw is rethrown.
// The ranges of one call may run in any order and not all at once: they must not wait for one another.  More generally, nothing that runs on the pool may
// wait for anything but its own nested parallel_ranges: a waiting thread helps with whatever is queued, so the task it waits for may be the one further
// down its own stack.  For the same reason no lock that a queued task could want is held across a call.
template <class F> void parallel_ranges(size_t n, size_t min_per_part, F&& f) {
    const size_t parts = std::max<size_t>(1, std::min<size_t>(host_threads(), min_per_part ? n / min_per_part : n));
    if (parts <= 1 || n == 0) { f((size_t)0, n, (size_t)0); return; }
    std::vector<std::exception_ptr> err(parts);
    struct Group { std::atomic<size_t> left; std::mutex m; std::condition_variable cv; } grp;
    grp.left.store(parts - 1);
    auto run = [&](size_t p) {
        const size_t b = n * p / parts, e = n * (p + 1) / parts;
        try { f(b, e, p); } catch (...) { err[p] = std::current_exception(); }
    };
    HostPool& pool = HostPool::get();
    for (size_t p = 1; p < parts; p++)
        pool.submit([&run, &grp, p] {
            run(p);
            std::lock_guard<std::mutex> g(grp.m);                                  // (under the lock: grp lives on the waiter's stack until it sees 0)
            if (grp.left.fetch_sub(1) == 1) grp.cv.notify_all();
        });
    run(0);
    while (grp.left.load() != 0) {
        if (pool.run_one()) continue;
        std::unique_lock<std::mutex> lk(grp.m);
        if (grp.left.load() != 0) grp.cv.wait_for(lk, std::chrono::microseconds(200));
    }
    { std::lock_guard<std::mutex> g(grp.m); }                                      // the last range has left its critical section
    for (auto& e : err) if (e) std::rethrow_exception(e);
}

}  // namespace rrt
