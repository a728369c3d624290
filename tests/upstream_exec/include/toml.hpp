// tests/upstream_exec (README.md there): the part of toml11 that /root/reference/src/iyokan.hpp:1731-1895 uses (NetworkBlueprint),
// WORKING, on top of this repository's own TOML reader (iyokan_amd/host/toml.hpp): parse(file), find<T>, find_or<T>, get<T> for
// std::string, integers, std::vector<std::string>, std::vector<value> and table.  Test infrastructure only; toml11 is not here.
#pragma once
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../../iyokan_amd/host/toml.hpp"

namespace toml {
class value;
typedef std::unordered_map<std::string, value> table;
typedef std::vector<value> array;

class value {
public:
    iyk::host::toml::Value v;
    value() {}
    explicit value(iyk::host::toml::Value x) : v(std::move(x)) {}
    bool is_array() const { return v.kind == iyk::host::toml::Value::Array; }
    bool is_string() const { return v.kind == iyk::host::toml::Value::String; }
    bool is_table() const { return v.kind == iyk::host::toml::Value::Table; }
    bool contains(const std::string& key) const { return v.find(key) != nullptr; }
};

inline value parse(const std::string& fileName)
{
    std::ifstream ifs{fileName};
    if (!ifs)
        throw std::runtime_error("toml stand-in: cannot open " + fileName);
    std::stringstream ss;
    ss << ifs.rdbuf();
    return value{iyk::host::toml::parse(ss.str())};
}

template <class T>
T get(const value& x)
{
    using V = iyk::host::toml::Value;
    if constexpr (std::is_same_v<T, std::string>) return x.v.asString();
    else if constexpr (std::is_integral_v<T>) return static_cast<T>(x.v.asInt());
    else if constexpr (std::is_same_v<T, std::vector<std::string>>) {
        std::vector<std::string> out;
        for (auto&& e : x.v.asArray()) out.push_back(e.asString());
        return out;
    }
    else if constexpr (std::is_same_v<T, std::vector<value>>) {
        std::vector<value> out;
        for (auto&& e : x.v.asArray()) out.emplace_back(e);
        return out;
    }
    else if constexpr (std::is_same_v<T, table>) {
        if (x.v.kind != V::Table)
            throw std::runtime_error("toml stand-in: table expected");
        table out;
        for (auto&& kv : x.v.tbl) out.emplace(kv.first, value{kv.second});
        return out;
    }
    else static_assert(sizeof(T) == 0, "toml stand-in: type not modelled");
}
template <class T>
T find(const value& x, const std::string& key)
{
    const iyk::host::toml::Value* p = x.v.find(key);
    if (!p)
        throw std::out_of_range("toml stand-in: key not found: " + key);
    return get<T>(value{*p});
}
template <class T>
T find_or(const value& x, const std::string& key, T&& fallback)
{
    const iyk::host::toml::Value* p = x.v.find(key);
    if (!p)
        return std::forward<T>(fallback);
    return get<std::remove_cv_t<std::remove_reference_t<T>>>(value{*p});
}
template <class T>
T find_or(const value& x, const std::string& key, const T& fallback)
{
    const iyk::host::toml::Value* p = x.v.find(key);
    return p ? get<T>(value{*p}) : fallback;
}
}  // namespace toml
