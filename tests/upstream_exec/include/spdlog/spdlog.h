// tests/upstream_exec (README.md there): spdlog's free functions as print statements (warnings and errors to stderr, the rest
// dropped), so that error::die() of /root/reference/src/error.hpp:22-48 says why it died.  Test infrastructure only.
#pragma once
#include <iostream>
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
template <class... A> void warn(const std::string& f, A&&... a) { std::cerr << "[warn] " << fmt::format(f, a...) << std::endl; }
template <class... A> void error(const std::string& f, A&&... a) { std::cerr << "[error] " << fmt::format(f, a...) << std::endl; }
}  // namespace spdlog
