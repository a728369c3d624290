// tests/upstream_exec (README.md there): a WORKING thread pool with the interface /root/reference/src/iyokan.hpp:1309-1346 uses
// (ThreadPool(size_t), enqueue(f) -> std::future).  Written here from that call site; test infrastructure only.
#pragma once
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

class ThreadPool {
private:
    std::vector<std::thread> threads_;
    std::deque<std::function<void()>> jobs_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;

    void loop()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty())
                    return;  // stop_ and drained
                job = std::move(jobs_.front());
                jobs_.pop_front();
            }
            job();
        }
    }

public:
    explicit ThreadPool(size_t n)
    {
        if (n == 0)
            n = 1;
        for (size_t i = 0; i < n; i++)
            threads_.emplace_back([this] { loop(); });
    }

    ~ThreadPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto&& t : threads_)
            t.join();
    }

    ThreadPool(const ThreadPool&) = delete;
    ThreadPool& operator=(const ThreadPool&) = delete;

    template <class F, class... Args>
    auto enqueue(F&& f, Args&&... args) -> std::future<std::invoke_result_t<F, Args...>>
    {
        using R = std::invoke_result_t<F, Args...>;
        auto task = std::make_shared<std::packaged_task<R()>>(std::bind(std::forward<F>(f), std::forward<Args>(args)...));
        std::future<R> fut = task->get_future();
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.emplace_back([task] { (*task)(); });
        }
        cv_.notify_one();
        return fut;
    }
};
