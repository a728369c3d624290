// compile-only shim (tests/shims/README.md): progschj/ThreadPool as /root/reference/src/iyokan.hpp:1309-1346 uses it
#pragma once
#include <cstddef>
#include <future>
#include <type_traits>
class ThreadPool {
public:
    explicit ThreadPool(size_t) {}
    template <class F, class... Args>
    auto enqueue(F&&, Args&&...) -> std::future<std::invoke_result_t<F, Args...>>
    {
        return {};   // declaration-level stand-in: nothing runs
    }
};
