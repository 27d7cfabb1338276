// parallel.hpp -- the host side's only threading primitive: a blocking parallel_for over [0, n) on std::thread workers.
// Used by the once-per-scene set-up stages (parse, triangle boxes, own-list index); the frame path has no host threads.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <exception>
#include <thread>
#include <vector>

namespace rrt {

inline unsigned host_threads() {
    if (const char* e = std::getenv("RRT_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) return (unsigned)std::min(v, 256); }
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hc ? hc : 1u, 32u));
}

// f(begin, end, part) on `parts` contiguous ranges of [0, n); parts <= host_threads().  An exception of the EARLIEST range that threw is rethrown.
template <class F> void parallel_ranges(size_t n, size_t min_per_part, F&& f) {
    const size_t parts = std::max<size_t>(1, std::min<size_t>(host_threads(), min_per_part ? n / min_per_part : n));
    if (parts <= 1 || n == 0) { f((size_t)0, n, (size_t)0); return; }
    std::vector<std::exception_ptr> err(parts);
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    auto run = [&](size_t p) {
        const size_t b = n * p / parts, e = n * (p + 1) / parts;
        try { f(b, e, p); } catch (...) { err[p] = std::current_exception(); }
    };
    for (size_t p = 1; p < parts; p++) th.emplace_back(run, p);
    run(0);
    for (auto& t : th) t.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
}

}  // namespace rrt
