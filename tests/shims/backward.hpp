// compile-only shim (tests/shims/README.md): what /root/reference/src/error.hpp:8,25-44 and error.cpp use of backward-cpp
#pragma once
#include <ostream>
namespace backward {
struct StackTrace {
    void load_here(int) {}
};
struct Printer {
    template <class S>
    void print(StackTrace&, S&) {}
};
struct SignalHandling {};
}  // namespace backward
