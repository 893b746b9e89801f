// worker_pool.h — the worker threads behind every multi-threaded host loop of this library (csr2cpsr and the channel assembly,
// data_formatter.h / channel_packets.h; the load-time builders, hisparse_amd/csrc/tiles_common.h).
#ifndef HISPARSE_WORKER_POOL_H_
#define HISPARSE_WORKER_POOL_H_

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace hisparse {

// The library's worker threads: started once (first parallel_for of the process), parked on a condition variable between jobs.
// Starting and joining 255 std::threads per loop cost 5-8 ms on the 256-thread host of the GPU box -- more than most of the loops
// they ran (the BITMAP builder has six of them).  One job at a time: a second caller (another context loading on another
// thread) or a task that calls parallel_for itself gets `false` and runs the loop with threads of its own.  The caller
// works through the indices as well, so a job completes even when the workers are gone (a forked child).
class WorkerPool {
  public:
    static WorkerPool& get() {
        static WorkerPool pool;
        return pool;
    }
    // at most `threads` threads work on the job, the caller included
    bool run(size_t n, unsigned threads, void (*call)(void*, size_t), void* ctx) {
        if (in_task()) return false;
        std::unique_lock<std::mutex> lk(m_);
        if (busy_) return false;
        busy_ = true;
        if (!started_) {
            started_ = true;
            const unsigned hw = std::thread::hardware_concurrency();
            for (unsigned t = 1; t < std::min(hw, 256u); ++t) threads_.emplace_back([this]() { worker(); });
        }
        idle_.wait(lk, [&]() { return active_ == 0; });
        call_ = call; ctx_ = ctx; n_ = n; finished_ = 0; failure_ = nullptr;
        seats_ = std::min<size_t>(threads, n) - 1;
        next_.store(0);
        ++generation_;
        lk.unlock();
        wake_.notify_all();
        const size_t mine = drain(n, call, ctx);
        lk.lock();
        finished_ += mine;
        idle_.wait(lk, [&]() { return finished_ == n_ && active_ == 0; });
        call_ = nullptr; n_ = 0; busy_ = false;
        std::exception_ptr failure = failure_;
        failure_ = nullptr;
        lk.unlock();
        if (failure) std::rethrow_exception(failure);
        return true;
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto& th : threads_) th.join();
    }

  private:
    static bool& in_task() {
        static thread_local bool flag = false;
        return flag;
    }
    // takes indices until they run out; an exception of a task is kept (the first one) and the remaining indices are still counted
    size_t drain(size_t n, void (*call)(void*, size_t), void* ctx) {
        size_t done = 0;
        in_task() = true;
        for (size_t i = next_.fetch_add(1); i < n; i = next_.fetch_add(1), ++done) {
            try {
                call(ctx, i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(failure_m_);
                if (!failure_) failure_ = std::current_exception();
            }
        }
        in_task() = false;
        return done;
    }
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            wake_.wait(lk, [&]() { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            if (!call_ || seats_ == 0) continue;   // woke up after the job was over, or the job has all the threads it wants
            --seats_;
            void (*call)(void*, size_t) = call_;
            void* ctx = ctx_;
            const size_t n = n_;
            ++active_;
            lk.unlock();
            const size_t mine = drain(n, call, ctx);
            lk.lock();
            finished_ += mine;
            if (--active_ == 0) idle_.notify_all();
        }
    }
    std::mutex m_, failure_m_;
    std::condition_variable wake_, idle_;
    std::vector<std::thread> threads_;
    bool started_ = false, stop_ = false, busy_ = false;
    uint64_t generation_ = 0;
    void (*call_)(void*, size_t) = nullptr;
    void* ctx_ = nullptr;
    size_t n_ = 0, finished_ = 0, seats_ = 0;
    unsigned active_ = 0;
    std::atomic<size_t> next_{0};
    std::exception_ptr failure_;
};

// fn(i) for every i in [0, n) on at most `threads` threads (the caller is one of them); returns when all are done; the first
// exception a task throws is rethrown here
template <typename Fn>
void pooled_for(size_t n, unsigned threads, Fn& fn) {
    threads = unsigned(std::min<size_t>(std::max(1u, threads), n));
    if (threads <= 1) {
        for (size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    if (WorkerPool::get().run(n, threads, [](void* f, size_t i) { (*static_cast<Fn*>(f))(i); }, &fn)) return;
    // The pool is busy (another context is loading on another thread) or this is a loop inside a task: threads of this call's own --
    // at most kFallbackThreads of them (the pool already keeps every core busy), the caller included, with the same exception
    // contract as the pooled path: the first exception a task throws is kept, the remaining indices still run, it is rethrown here.
    constexpr unsigned kFallbackThreads = 16;
    threads = std::min(threads, kFallbackThreads);
    std::atomic<size_t> next(0);
    std::mutex failure_m;
    std::exception_ptr failure;
    auto work = [&]() {
        for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            try {
                fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(failure_m);
                if (!failure) failure = std::current_exception();
            }
        }
    };
    std::vector<std::thread> own;
    own.reserve(threads - 1);
    for (unsigned t = 1; t < threads; ++t) {
        try {
            own.emplace_back(work);
        } catch (...) {      // no more threads to be had (std::system_error): the ones there are, and the caller, do the work
            break;
        }
    }
    work();
    for (auto& th : own) th.join();
    if (failure) std::rethrow_exception(failure);
}

}  // namespace hisparse

#endif  // HISPARSE_WORKER_POOL_H_
