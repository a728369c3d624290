// compile-only shim (tests/shims/README.md): the part of toml11 /root/reference/src uses (iyokan.hpp:1731-1895,
// iyokan-packet.cpp is not compiled here)
#pragma once
#include <string>
#include <unordered_map>
#include <vector>
namespace toml {
class value;
typedef std::unordered_map<std::string, value> table;
typedef std::vector<value> array;
class value {
public:
    value();
    value(const value&);
    value& operator=(const value&);
    ~value();
    bool is_array() const;
    bool is_string() const;
    bool is_table() const;
    bool contains(const std::string&) const;
};
value parse(const std::string& fileName);
template <class T> T find(const value&, const std::string& key);
template <class T> T find_or(const value&, const std::string& key, T&& fallback);
template <class T> T find_or(const value&, const std::string& key, const T& fallback);
template <class T> T get(const value&);
}  // namespace toml
