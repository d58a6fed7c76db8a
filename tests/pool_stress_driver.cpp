// Stress of the search's fork-join pool (batch_first_amd/csrc/search/threads.h) for tests/test_search_sanitizers.py:
// many short jobs of changing size on more threads than the box has cores to spare, narrow and wide ones mixed; every
// index of every job must be visited exactly once (a piece claimed twice, or a piece of the NEXT job claimed with
// this job's description, shows up as a count other than 1 or as a job that never ends: the test has a timeout).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "threads.h"

int main(int argc, char** argv) {
    const int threads = argc > 1 ? std::atoi(argv[1]) : 8, jobs = argc > 2 ? std::atoi(argv[2]) : 200000;
    bfs::Pool pool(threads);
    std::vector<std::atomic<int>> hits(1 << 14);
    unsigned long long x = 88172645463325252ull, bad = 0, items = 0;
    for (int j = 0; j < jobs; ++j) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const size_t n = 1 + (size_t)(x % ((j % 7 == 0) ? hits.size() : 300));   // mostly small jobs, some large
        const size_t grain = 1 + (size_t)((x >> 20) % 97);
        for (size_t i = 0; i < n; ++i) hits[i].store(0, std::memory_order_relaxed);
        pool.for_range(n, grain, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) hits[i].fetch_add(1, std::memory_order_relaxed);
        }, (x >> 40) & 1);
        for (size_t i = 0; i < n; ++i) bad += hits[i].load(std::memory_order_relaxed) != 1;
        items += n;
    }
    std::printf("jobs %d items %llu bad %llu\n", jobs, items, bad);
    return bad ? 1 : 0;
}
