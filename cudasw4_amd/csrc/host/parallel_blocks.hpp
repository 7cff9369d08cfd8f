// parallel_blocks.hpp — fork-join helper for the few parallel loops of the host library (DB validation, the pinned
// staging copy of streamed shards).  Deliberately NOT OpenMP: libgomp's worker threads busy-wait after a parallel region,
// and inside a container with a CPU quota that spinning burns the quota — the whole process, including the thread that
// feeds the GPU, is then throttled for most of the next scheduler periods.  Measured (tools/evict_probe.py,
// tools/cold_start_probe.py): two or three 70-80 ms stalls 100 ms apart after every DB load, i.e. inside the first
// queries of a cold `align` run; none with OMP_NUM_THREADS=1, none with these threads, which exit when the loop is done.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace swh {

// fn(i) for every i in [0, n), blocks handed out dynamically to at most max_threads threads (the caller is one of them);
// the first exception thrown by fn is rethrown in the caller after all threads have joined
template <class F>
void parallel_blocks(size_t n, unsigned max_threads, F&& fn) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = unsigned(std::min<size_t>(std::min(max_threads, hw), n));
    if (nt <= 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::exception_ptr error;
    std::mutex error_mutex;
    auto work = [&]() {
        try {
            for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
        } catch (...) {
            std::lock_guard<std::mutex> lock(error_mutex);
            if (!error) error = std::current_exception();
            next.store(n);
        }
    };
    std::vector<std::thread> threads;
    threads.reserve(nt - 1);
    // thread creation can fail (a container's pids limit, RLIMIT_NPROC): the threads that did start, and the caller,
    // finish the loop — an exception here would destroy joinable threads and terminate the process
    for (unsigned t = 1; t < nt; t++) {
        try {
            threads.emplace_back(work);
        } catch (const std::system_error&) {
            break;
        }
    }
    work();
    for (auto& t : threads) t.join();
    if (error) std::rethrow_exception(error);
}

}  // namespace swh
