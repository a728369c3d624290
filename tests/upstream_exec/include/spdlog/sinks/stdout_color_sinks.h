#pragma once
#include "../spdlog.h"
namespace spdlog {
inline std::shared_ptr<logger> stderr_color_mt(const std::string&) { return nullptr; }
}
