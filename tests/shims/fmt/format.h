// compile-only shim (tests/shims/README.md): fmt::format / fmt::fprintf as /root/reference/src uses them
#pragma once
#include <ostream>
#include <string>
namespace fmt {
template <class... A>
std::string format(const char*, A&&...) { return {}; }
template <class... A>
std::string format(const std::string&, A&&...) { return {}; }
template <class... A>
int fprintf(std::ostream&, const char*, A&&...) { return 0; }
template <class... A>
std::string sprintf(const char*, A&&...) { return {}; }
}  // namespace fmt
