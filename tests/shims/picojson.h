// compile-only shim (tests/shims/README.md): the part of picojson /root/reference/src/iyokan.hpp uses (:229-257, 2115-2482)
#pragma once
#include <istream>
#include <map>
#include <ostream>
#include <string>
#include <vector>
namespace picojson {
class value;
typedef std::vector<value> array;
typedef std::map<std::string, value> object;
class value {
public:
    value();
    value(double);
    value(bool);
    value(const std::string&);
    value(const char*);
    value(const array&);
    value(const object&);
    value(const value&);
    value& operator=(const value&);
    ~value();
    template <class T> bool is() const;
    template <class T> const T& get() const;
    template <class T> T& get();
    bool contains(const std::string&) const;
    std::string serialize(bool prettify = false) const;
    std::string to_str() const;
};
std::string parse(value&, std::istream&);
std::string parse(value&, const std::string&);
std::ostream& operator<<(std::ostream&, const value&);
std::istream& operator>>(std::istream&, value&);
}  // namespace picojson
