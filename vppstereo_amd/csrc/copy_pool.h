// copy_pool.h -- a few threads that copy host memory: the frame stream's staging copies (vppx_fstream.hip).
// Plain C++17 (no HIP): tests/test_copy_pool_cpu.py builds it with g++ -fsanitize=thread and hammers it.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace vppx_host {

// one frame is 5-7 MB, one core moves 8-20 GB/s
struct CopyJob { void *dst; const void *src; size_t bytes; };
class CopyPool {
public:
    explicit CopyPool(int nthreads)
    {
        for (int i = 0; i < nthreads; i++) workers_.emplace_back([this] { run(); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    // copies every job, cut into pieces; the caller's thread works too; returns when everything is copied
    void copy(const CopyJob *jobs, int n)
    {
        const size_t piece = 256 * 1024;
        pieces_.clear();
        for (int i = 0; i < n; i++) {
            if (!jobs[i].dst || !jobs[i].src) continue;
            for (size_t o = 0; o < jobs[i].bytes; o += piece)
                pieces_.push_back({(char *)jobs[i].dst + o, (const char *)jobs[i].src + o, jobs[i].bytes - o < piece ? jobs[i].bytes - o : piece});
        }
        if (pieces_.empty()) return;
        if (workers_.empty() || pieces_.size() == 1) {
            for (auto &p : pieces_) memcpy(p.dst, p.src, p.bytes);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            next_.store(0);
            left_.store((int)pieces_.size());
            open_ = true;
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        work();
        // the other threads' last pieces: a few microseconds, not worth a sleep
        for (int spin = 0; spin < 100000 && (left_.load(std::memory_order_acquire) != 0 || active_.load(std::memory_order_acquire) != 0); spin++)
            __builtin_ia32_pause();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return left_.load() == 0 && active_.load() == 0; });
        open_ = false; // (workers that wake up late find nothing to join)
    }

private:
    void work()
    {
        int mine = 0;
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= pieces_.size()) break;
            memcpy(pieces_[i].dst, pieces_[i].src, pieces_[i].bytes);
            mine++;
        }
        if (mine) left_.fetch_sub(mine, std::memory_order_release);
    }
    void run()
    {
        unsigned seen = 0;
        for (;;) {
            // frames of a running stream follow each other within a few hundred microseconds: look for the next one for a while
            // before going to sleep (a condition variable's wake-up costs as much as the copy it is woken for)
            for (int spin = 0; spin < 20000 && gen_.load(std::memory_order_acquire) == seen; spin++) __builtin_ia32_pause();
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_.load() != seen; });
                if (stop_) return;
                seen = gen_.load();
                if (!open_) continue; // that copy is over already
                active_.fetch_add(1);
            }
            work();
            {
                std::lock_guard<std::mutex> lk(m_);
                active_.fetch_sub(1, std::memory_order_release);
            }
            done_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<CopyJob> pieces_;
    std::atomic<size_t> next_{0};
    std::atomic<int> left_{0}, active_{0};
    std::atomic<unsigned> gen_{0};
    bool open_ = false, stop_ = false;
};

} // namespace vppx_host
