// The load-time builders' worker pool (hisparse_amd/csrc/tiles_common.h: parallel_for): every index exactly once, concurrent callers,
// nested calls, an exception of a task reaches the caller and the pool is usable afterwards.
#undef NDEBUG
#include "tiles_common.h"
#include <cassert>
#include <iostream>
#include <stdexcept>
using namespace hisparse::dev::detail;
int main() {
    // sequential jobs of varying size
    for (int round = 0; round < 2000; ++round) {
        size_t n = 1 + (round * 7919) % 300;
        std::vector<int> hit(n, 0);
        parallel_for(n, [&](size_t i) { hit[i]++; });
        for (size_t i = 0; i < n; ++i) assert(hit[i] == 1);
    }
    // concurrent callers
    std::vector<std::thread> callers;
    std::atomic<long> total(0);
    for (int c = 0; c < 6; ++c)
        callers.emplace_back([&, c]() {
            for (int round = 0; round < 300; ++round) {
                size_t n = 2 + (round * 31 + c) % 97;
                std::vector<int> hit(n, 0);
                parallel_for(n, [&](size_t i) { hit[i]++; });
                for (size_t i = 0; i < n; ++i) assert(hit[i] == 1);
                total += long(n);
            }
        });
    for (auto& t : callers) t.join();
    // a capped job uses at most that many threads (HISPARSE_FORMAT_THREADS)
    {
        std::mutex m;
        std::vector<std::thread::id> seen;
        auto note = [&](size_t) {
            std::lock_guard<std::mutex> lk(m);
            if (std::find(seen.begin(), seen.end(), std::this_thread::get_id()) == seen.end()) seen.push_back(std::this_thread::get_id());
        };
        for (int round = 0; round < 50; ++round) {
            seen.clear();
            hisparse::pooled_for(1000, 3, note);
            assert(seen.size() <= 3);
        }
    }
    // nested
    std::atomic<int> inner(0);
    parallel_for(8, [&](size_t) { parallel_for(5, [&](size_t) { inner++; }); });
    assert(inner == 40);
    // exception
    bool caught = false;
    try {
        parallel_for(64, [&](size_t i) { if (i == 13) throw std::bad_alloc(); });
    } catch (const std::bad_alloc&) { caught = true; }
    assert(caught);
    // ... also on the fallback path (a loop inside a task runs on threads of its own): the inner exception comes out of the inner
    // call, every other inner index still runs, and the outer job carries it to its caller
    std::atomic<int> ran(0);
    caught = false;
    try {
        parallel_for(4, [&](size_t) {
            parallel_for(40, [&](size_t i) { ran++; if (i == 7) throw std::length_error("inner"); });
        });
    } catch (const std::length_error&) { caught = true; }
    assert(caught && ran == 160);
    std::vector<int> hit(100, 0);
    parallel_for(100, [&](size_t i) { hit[i]++; });
    for (int v : hit) assert(v == 1);
    std::cout << "WORKER POOL OK " << total << "\n";
}
