// Host threads of the search driver: a fork-join pool for the per-node work of one batch (child
// construction, draw / table / mate checks, move generation, record packing: the loops the reference
// runs under numba prange, numba_board.py:242-266) and a lane that runs evaluator calls beside it.
//
// An iteration of the search lasts tens of microseconds to a few milliseconds, so waiting threads spin
// for a short while before they sleep: a sleeping thread costs a futex wake (20-50 us) per hand-over,
// which is as long as a small iteration.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace bfs {

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

// A generation counter that waiters watch: spin for `spin_us`, then sleep on the condition variable.
class Gate {
public:
    explicit Gate(int spin_us = 200) : spin_us_(spin_us) {}
    uint64_t value() const { return gen_.load(); }
    void bump() {
        gen_.fetch_add(1);
        if (sleepers_.load() > 0) {
            std::lock_guard<std::mutex> lk(m_);
            cv_.notify_all();
        }
    }
    // returns once the counter differs from `seen`
    void wait_change(uint64_t seen) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0;; ++i) {
            if (gen_.load() != seen) return;
            cpu_relax();
            if ((i & 63) == 63 &&
                std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us_)
                break;
        }
        std::unique_lock<std::mutex> lk(m_);
        sleepers_.fetch_add(1);
        cv_.wait(lk, [&] { return gen_.load() != seen; });
        sleepers_.fetch_sub(1);
    }

private:
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> sleepers_{0};
    std::mutex m_;
    std::condition_variable cv_;
    int spin_us_;
};

// Fork-join over an index range; the calling thread takes part.  One job at a time.
// The workers form two groups: the first `narrow - 1` of them take part in every job, the others only in "wide"
// ones.  A search at shallow depth hands over a hundred nodes forty thousand times a second: a dozen workers that
// all spin for those jobs gain nothing on three and take the cores that the evaluator's threads need (depth 4:
// 3.2 M nodes/s with twelve threads, 3.7 M with two); while only narrow jobs come, the second group goes to sleep.
//
// A job ends when all its ITEMS are done, not when every worker has reported: a worker that is asleep or that the
// operating system has taken off its core (the box runs the evaluators' threads and the GPU runtime's beside the
// pool) arrives late, finds nothing left and goes back to waiting, and nobody waited for it.  (Round 2's pool counted
// workers: on a busy box every hand-over then cost the slowest worker's wake-up, and the open list's validity check
// took four times as long on eight threads as on one.)  Pieces are claimed by compare-and-swap on a word that
// carries the job's number, and jobs alternate between two descriptions: a late worker that still holds job g's
// word reads description g & 1, which nobody rewrites before job g + 1 has been published (after which its
// compare-and-swap fails); while the word still says "job g, everything claimed" the description is intact and tells
// it that nothing is left.  (With ONE description the worker could read the next job's larger item count beside the
// old word and claim a piece that does not exist yet: the first version of this pool hung that way.)
class Pool {
public:
    // threads = total including the caller; workers that cannot be started are done without
    explicit Pool(int threads, int narrow = 3) {
        for (int i = 1; i < threads; ++i) {
            const int group = i < narrow ? 0 : 1;
            try {
                workers_.emplace_back([this, group] { worker(group); });
            } catch (...) {
                break;
            }
            ++in_group_[group];
        }
    }
    ~Pool() {
        stop_.store(true);
        start_[0].bump();
        start_[1].bump();
        for (auto& t : workers_) t.join();
    }
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;
    int threads() const { return (int)workers_.size() + 1; }
    int narrow_threads() const { return in_group_[0] + 1; }

    // f(begin, end) over [0, n) in pieces of `grain`; f must not throw
    template <class F>
    void for_range(size_t n, size_t grain, F&& f, bool wide = true) {
        if (workers_.empty() || n <= grain) {
            if (n) f((size_t)0, n);
            return;
        }
        struct Ctx { F* f; } ctx{&f};
        // the job's description is written before its number is published and stays untouched until its last item
        // is done; a piece can only be claimed while the job is unfinished (see run_pieces)
        const uint64_t job = ++job_;
        Desc& d = desc_[job & 1];
        d.call.store([](void* c, size_t b, size_t e) { (*static_cast<Ctx*>(c)->f)(b, e); }, std::memory_order_relaxed);
        d.ctx.store(&ctx, std::memory_order_relaxed);
        d.total.store(n, std::memory_order_relaxed);
        d.grain.store(grain, std::memory_order_relaxed);
        done_.store(0, std::memory_order_relaxed);
        next_.store(job << 32, std::memory_order_release);
        start_[0].bump();
        if (wide && in_group_[1] > 0) start_[1].bump();
        run_pieces();
        while (done_.load(std::memory_order_acquire) != n) cpu_relax();
    }

private:
    void run_pieces() {
        for (;;) {
            uint64_t cur = next_.load(std::memory_order_acquire);
            // (relaxed atomics: a worker that is two jobs late may read these while the caller rewrites them; its
            // claim below then fails, and what it read is dropped)
            const Desc& d = desc_[(cur >> 32) & 1];
            const size_t total = d.total.load(std::memory_order_relaxed), grain = d.grain.load(std::memory_order_relaxed);
            const CallFn call = d.call.load(std::memory_order_relaxed);
            void* const ctx = d.ctx.load(std::memory_order_relaxed);
            const size_t b = (size_t)(cur & 0xFFFFFFFFull);
            if (b >= total) return;
            const size_t e = b + grain < total ? b + grain : total;
            // succeeds only if nobody has claimed [b, e) of THIS job: the job is then unfinished, so what was read
            // above is its description
            if (!next_.compare_exchange_weak(cur, (cur & ~0xFFFFFFFFull) | e, std::memory_order_acq_rel)) continue;
            call(ctx, b, e);
            done_.fetch_add(e - b, std::memory_order_release);
        }
    }
    void worker(int group) {
        uint64_t seen = 0;
        for (;;) {
            start_[group].wait_change(seen);
            seen = start_[group].value();
            if (stop_.load()) return;
            run_pieces();
        }
    }

    std::vector<std::thread> workers_;
    int in_group_[2] = {0, 0};
    Gate start_[2];
    std::atomic<bool> stop_{false};
    std::atomic<uint64_t> next_{0};   // job number << 32 | first unclaimed index
    std::atomic<size_t> done_{0};     // items of the current job that are finished
    uint64_t job_ = 0;
    using CallFn = void (*)(void*, size_t, size_t);
    struct Desc {
        std::atomic<size_t> total{0}, grain{1};
        std::atomic<CallFn> call{nullptr};
        std::atomic<void*> ctx{nullptr};
    };
    Desc desc_[2];
};

// One thread that runs the calls handed to it, in order; wait() returns when all of them are done.
class Lane {
public:
    using Fn = int (*)(void* arg);
    Lane() {
        try {
            thread_ = std::thread([this] { run(); });
            ok_ = true;
        } catch (...) {
            ok_ = false;
        }
    }
    ~Lane() {
        if (!ok_) return;
        stop_.store(true);
        work_.bump();
        thread_.join();
    }
    Lane(const Lane&) = delete;
    Lane& operator=(const Lane&) = delete;
    bool ok() const { return ok_; }

    void submit(Fn fn, void* arg) {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back({fn, arg});
        }
        submitted_.fetch_add(1);
        work_.bump();
    }
    // nonzero if any call since the last wait() failed
    int wait() {
        while (done_.load() != submitted_.load()) cpu_relax();
        return rc_.exchange(0);
    }

private:
    struct Job { Fn fn; void* arg; };
    void run() {
        uint64_t seen = 0;
        for (;;) {
            for (;;) {
                Job j;
                {
                    std::lock_guard<std::mutex> lk(m_);
                    if (q_.empty()) break;
                    j = q_.front();
                    q_.pop_front();
                }
                const int rc = j.fn(j.arg);
                if (rc) rc_.store(rc);
                done_.fetch_add(1);
            }
            if (stop_.load()) return;
            work_.wait_change(seen);
            seen = work_.value();
        }
    }

    std::thread thread_;
    bool ok_ = false;
    Gate work_;
    std::atomic<bool> stop_{false};
    std::atomic<uint64_t> submitted_{0}, done_{0};
    std::atomic<int> rc_{0};
    std::mutex m_;
    std::deque<Job> q_;
};

}  // namespace bfs
