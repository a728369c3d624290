// tests/upstream_exec (README.md there): fmt::format / sprintf / fprintf that actually format, for the handful of call sites in
// /root/reference/src (utility.hpp:34, test0.cpp:500-519, spdlog messages).  "{}" placeholders and printf conversions of
// integers, floating point, strings; nothing else of fmt is modelled.  Test infrastructure only.
#pragma once
#include <cctype>
#include <cstdio>
#include <ostream>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace fmt {
namespace exec_detail {
template <class T>
std::string show(const T& v)
{
    std::ostringstream ss;
    ss << v;
    return ss.str();
}

inline std::string braces(const std::string& f, const std::vector<std::string>& args)
{
    std::string out;
    size_t next = 0;
    for (size_t i = 0; i < f.size(); i++) {
        if (f[i] == '{' && i + 1 < f.size() && f[i + 1] == '{') {
            out += '{';
            i++;
        }
        else if (f[i] == '}' && i + 1 < f.size() && f[i + 1] == '}') {
            out += '}';
            i++;
        }
        else if (f[i] == '{') {
            const size_t close = f.find('}', i);
            if (close == std::string::npos)
                break;
            if (next < args.size())
                out += args[next++];
            i = close;
        }
        else
            out += f[i];
    }
    return out;
}

template <class T>
std::string conv(const std::string& spec, const T& v)
{
    char buf[512];
    const char kind = spec.back();
    if constexpr (std::is_arithmetic_v<T>) {
        std::string s = spec.substr(0, spec.size() - 1);
        while (!s.empty() && (s.back() == 'l' || s.back() == 'h' || s.back() == 'z'))
            s.pop_back();
        if (kind == 'd' || kind == 'i') {
            std::snprintf(buf, sizeof buf, (s + "lld").c_str(), static_cast<long long>(v));
            return buf;
        }
        if (kind == 'u' || kind == 'x' || kind == 'X' || kind == 'o') {
            std::snprintf(buf, sizeof buf, (s + "ll" + kind).c_str(), static_cast<unsigned long long>(v));
            return buf;
        }
        if (kind == 'f' || kind == 'e' || kind == 'g') {
            std::snprintf(buf, sizeof buf, (s + kind).c_str(), static_cast<double>(v));
            return buf;
        }
        if (kind == 'c') {
            return std::string(1, static_cast<char>(v));
        }
    }
    return show(v);
}

inline void walk(std::string& out, const char*& f)
{
    for (; *f; f++) {
        if (f[0] == '%' && f[1] == '%') {
            out += '%';
            f++;
        }
        else if (f[0] == '%')
            return;
        else
            out += *f;
    }
}

inline std::string takeSpec(const char*& f)
{
    std::string spec(1, *f++);
    while (*f && !std::isalpha(static_cast<unsigned char>(*f)))
        spec += *f++;
    while (*f == 'l' || *f == 'h' || *f == 'z')
        spec += *f++;
    if (*f)
        spec += *f++;
    return spec;
}

inline void printfInto(std::string& out, const char* f)
{
    walk(out, f);
}

template <class T, class... Rest>
void printfInto(std::string& out, const char* f, const T& v, const Rest&... rest)
{
    walk(out, f);
    if (!*f)
        return;
    const std::string spec = takeSpec(f);
    out += conv(spec, v);
    printfInto(out, f, rest...);
}
}  // namespace exec_detail

template <class... A>
std::string format(const std::string& f, const A&... a)
{
    return exec_detail::braces(f, {exec_detail::show(a)...});
}

template <class... A>
std::string sprintf(const char* f, const A&... a)
{
    std::string out;
    exec_detail::printfInto(out, f, a...);
    return out;
}

template <class... A>
int fprintf(std::ostream& os, const char* f, const A&... a)
{
    const std::string s = fmt::sprintf(f, a...);
    os << s;
    return static_cast<int>(s.size());
}
}  // namespace fmt
