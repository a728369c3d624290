// compile-only shim (tests/shims/README.md): the spdlog calls in /root/reference/src (error.hpp, iyokan_*.cpp, packet.hpp)
#pragma once
#include <memory>
#include <string>
#include "../fmt/format.h"
namespace spdlog {
namespace level {
enum level_enum { trace, debug, info, warn, err, critical, off };
}
class logger {};
inline void set_level(level::level_enum) {}
inline void drop_all() {}
inline void set_default_logger(std::shared_ptr<logger>) {}
template <class... A> void trace(const std::string&, A&&...) {}
template <class... A> void debug(const std::string&, A&&...) {}
template <class... A> void info(const std::string&, A&&...) {}
template <class... A> void warn(const std::string&, A&&...) {}
template <class... A> void error(const std::string&, A&&...) {}
}  // namespace spdlog
