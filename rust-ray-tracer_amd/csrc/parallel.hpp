// parallel.hpp -- the host side's only threading primitives: a blocking parallel_for over [0, n) and a single task beside the caller's own work.
// Used by the once-per-scene set-up stages (parse, texture decode, triangle boxes, own-list index, staging copies -- the copy-out of a frame rendered into a pageable host buffer included; tracing itself has no host threads).
//
// Work runs on a process-wide pool of workers that is grown on demand and never torn down: creating a thread costs 30-60 us and threads of one
// process are created one at a time (they share the address-space lock), so the ~500 short-lived threads of a six-texture scene load -- six decoders,
// each with four parallel stages -- cost more than the work they did.  What is queued on the pool are only OFFERS: "take unclaimed ranges of this call
// and run them".  The caller claims ranges of its own call in the same way until none is left and then waits for the ones in flight, so a call
// completes with or without workers (a forked child has none), nested calls cannot starve, and a waiting thread never runs anything but its own
// call's ranges -- a first version let waiters help with whatever was queued, and a waiter was promptly handed, further up its own stack, a task that
// waited for the one below it.
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
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace rrt {

// Threads a parallel stage may use: RRT_HOST_THREADS, else the hardware threads -- divided by the ranks of this node when the process is one of
// several (LOCAL_WORLD_SIZE, else WORLD_SIZE: one process per GPU, all loading the same scene at the same moment) -- and at most 128.
inline unsigned host_threads() {
    if (const char* e = std::getenv("RRT_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) return (unsigned)std::min(v, 256); }
    unsigned hc = std::thread::hardware_concurrency();
    const char* w = std::getenv("LOCAL_WORLD_SIZE");
    if (!w) w = std::getenv("WORLD_SIZE");
    if (w) { const int ranks = std::atoi(w); if (ranks > 1) hc = std::max(1u, hc / (unsigned)std::min(ranks, 64)); }
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
            catch (...) { std::lock_guard<std::mutex> g(m_); n_workers_--; }         // no thread to be had: whoever waits for the work does it himself
        }
        cv_.notify_one();
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
// The ranges of one call may run in any order and not all at once: they must not wait for one another.
template <class F> void parallel_ranges(size_t n, size_t min_per_part, F&& f) {
    const size_t parts = std::max<size_t>(1, std::min<size_t>(host_threads(), min_per_part ? n / min_per_part : n));
    if (parts <= 1 || n == 0) { f((size_t)0, n, (size_t)0); return; }
    std::vector<std::exception_ptr> err(parts);
    struct Group {
        std::atomic<size_t> next{0}, left{0}; size_t parts = 0;
        std::function<void(size_t)> run;                                          // refers to the caller's frame: called for claimed ranges only, all of which end before the caller returns
        std::mutex m; std::condition_variable cv;
        void drain() {
            for (size_t p; (p = next.fetch_add(1)) < parts;) {
                run(p);
                if (left.fetch_sub(1) == 1) { std::lock_guard<std::mutex> g(m); cv.notify_all(); }
            }
        }
    };
    auto grp = std::make_shared<Group>();
    grp->parts = parts; grp->left.store(parts);
    grp->run = [&](size_t p) {
        const size_t b = n * p / parts, e = n * (p + 1) / parts;
        try { f(b, e, p); } catch (...) { err[p] = std::current_exception(); }
    };
    HostPool& pool = HostPool::get();
    for (size_t p = 1; p < parts; p++) pool.submit([grp] { grp->drain(); });      // an offer that comes too late finds nothing to claim
    grp->drain();
    { std::unique_lock<std::mutex> lk(grp->m); grp->cv.wait(lk, [&] { return grp->left.load() == 0; }); }
    for (auto& e : err) if (e) std::rethrow_exception(e);
}

// One task beside the caller's own work: offered to the pool at construction; wait() runs it on the spot if no worker has taken it yet, otherwise waits
// for the worker, and rethrows what the task threw.  The destructor waits for a task in flight (it refers to the caller's frame) and drops one that has not started.  May be waited for from anywhere,
// also from a range running on the pool, as long as tasks do not wait for each other in a circle.
class AsyncTask {
    struct State { std::atomic<int> claimed{0}; std::function<void()> fn; bool done = false; std::exception_ptr err; std::mutex m; std::condition_variable cv;
        void run() { try { fn(); } catch (...) { err = std::current_exception(); } { std::lock_guard<std::mutex> g(m); done = true; } cv.notify_all(); } };
    std::shared_ptr<State> s_;
    void finish() { if (!s_->claimed.exchange(1)) s_->run(); else { std::unique_lock<std::mutex> lk(s_->m); s_->cv.wait(lk, [&] { return s_->done; }); } }
public:
    template <class F> explicit AsyncTask(F&& f) : s_(std::make_shared<State>()) {
        s_->fn = std::forward<F>(f);
        auto s = s_;
        HostPool::get().submit([s] { if (!s->claimed.exchange(1)) s->run(); });
    }
    AsyncTask(const AsyncTask&) = delete; AsyncTask& operator=(const AsyncTask&) = delete;
    void wait() { finish(); if (s_->err) { auto e = s_->err; s_->err = nullptr; std::rethrow_exception(e); } }
    ~AsyncTask() {                                                                 // a task nobody has started is dropped, one in flight is waited for
        if (s_->claimed.exchange(1)) { std::unique_lock<std::mutex> lk(s_->m); s_->cv.wait(lk, [&] { return s_->done; }); }
    }
};

}  // namespace rrt
