// CPU test of cudasw4_amd/csrc/host/parallel_blocks.hpp (compiled and run by tests/test_host_cpu.py)
#include "parallel_blocks.hpp"

#include <cstdio>
#include <numeric>
#include <stdexcept>

int main() {
    // every index exactly once, for block counts below / at / above the thread count, incl. 0 and 1
    for (size_t n : {size_t(0), size_t(1), size_t(3), size_t(8), size_t(1000)}) {
        for (unsigned threads : {1u, 4u, 64u}) {
            std::vector<std::atomic<int>> hits(n);
            for (auto& h : hits) h.store(0);
            swh::parallel_blocks(n, threads, [&](size_t i) { hits[i].fetch_add(1); });
            for (size_t i = 0; i < n; i++)
                if (hits[i].load() != 1) { std::printf("FAIL n=%zu threads=%u index %zu hit %d times\n", n, threads, i, hits[i].load()); return 1; }
        }
    }
    // an exception thrown in a worker reaches the caller after all threads have joined, and stops the hand-out
    std::atomic<size_t> done{0};
    bool caught = false;
    try {
        swh::parallel_blocks(100000, 8, [&](size_t i) {
            if (i == 17) throw std::runtime_error("block 17");
            done.fetch_add(1);
        });
    } catch (const std::runtime_error& e) {
        caught = std::string(e.what()) == "block 17";
    }
    if (!caught) { std::printf("FAIL exception not propagated\n"); return 1; }
    if (done.load() >= 100000 - 1) { std::printf("FAIL hand-out did not stop after the exception\n"); return 1; }
    std::printf("ok\n");
    return 0;
}
