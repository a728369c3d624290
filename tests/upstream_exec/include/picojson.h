// tests/upstream_exec (README.md there): a small WORKING JSON value + recursive-descent parser with the part of picojson's
// interface that /root/reference/src/iyokan.hpp uses (:229-257 writer, :2115-2352 Yosys reader, :2371-2482 Iyokan-L1 reader):
// value / array / object, is<T>(), get<T>(), contains(), parse(value&, istream&) -> error text, operator<< / >>.
// Written here from those call sites (picojson itself is not in this container); test infrastructure only.
#pragma once
#include <cassert>
#include <cstdlib>
#include <istream>
#include <iterator>
#include <map>
#include <memory>
#include <ostream>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace picojson {
struct null {
};
class value;
typedef std::vector<value> array;
typedef std::map<std::string, value> object;

class value {
private:
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind_;
    bool b_;
    double d_;
    std::unique_ptr<std::string> s_;
    std::unique_ptr<array> a_;
    std::unique_ptr<object> o_;

    void copyFrom(const value& x);

public:
    value() : kind_(NUL), b_(false), d_(0) {}
    value(double d) : kind_(NUM), b_(false), d_(d) {}
    value(bool b) : kind_(BOOL), b_(b), d_(0) {}
    value(const std::string& s) : kind_(STR), b_(false), d_(0), s_(new std::string(s)) {}
    value(const char* s) : kind_(STR), b_(false), d_(0), s_(new std::string(s)) {}
    value(const array& a);
    value(const object& o);
    value(const value& x) : kind_(NUL), b_(false), d_(0) { copyFrom(x); }
    value& operator=(const value& x)
    {
        if (this != &x)
            copyFrom(x);
        return *this;
    }
    ~value();

    template <class T> bool is() const;
    template <class T> const T& get() const;
    template <class T> T& get() { return const_cast<T&>(static_cast<const value*>(this)->get<T>()); }
    bool contains(const std::string& key) const;
    std::string serialize(bool prettify = false) const;
    std::string to_str() const { return kind_ == STR ? *s_ : serialize(); }
};

inline value::value(const array& a) : kind_(ARR), b_(false), d_(0), a_(new array(a)) {}
inline value::value(const object& o) : kind_(OBJ), b_(false), d_(0), o_(new object(o)) {}
inline value::~value() {}
inline void value::copyFrom(const value& x)
{
    kind_ = x.kind_;
    b_ = x.b_;
    d_ = x.d_;
    s_.reset(x.s_ ? new std::string(*x.s_) : nullptr);
    a_.reset(x.a_ ? new array(*x.a_) : nullptr);
    o_.reset(x.o_ ? new object(*x.o_) : nullptr);
}

template <class T>
bool value::is() const
{
    if constexpr (std::is_same_v<T, null>) return kind_ == NUL;
    else if constexpr (std::is_same_v<T, bool>) return kind_ == BOOL;
    else if constexpr (std::is_same_v<T, double>) return kind_ == NUM;
    else if constexpr (std::is_same_v<T, std::string>) return kind_ == STR;
    else if constexpr (std::is_same_v<T, array>) return kind_ == ARR;
    else if constexpr (std::is_same_v<T, object>) return kind_ == OBJ;
    else return false;
}

template <class T>
const T& value::get() const
{
    assert(is<T>() && "picojson::value::get<T>(): type mismatch");
    if constexpr (std::is_same_v<T, bool>) return b_;
    else if constexpr (std::is_same_v<T, double>) return d_;
    else if constexpr (std::is_same_v<T, std::string>) return *s_;
    else if constexpr (std::is_same_v<T, array>) return *a_;
    else return *o_;
}

inline bool value::contains(const std::string& key) const
{
    return kind_ == OBJ && o_->count(key) != 0;
}

namespace exec_detail {
inline void quote(std::ostream& os, const std::string& s)
{
    os << '"';
    for (char c : s) {
        switch (c) {
        case '"': os << "\\\""; break;
        case '\\': os << "\\\\"; break;
        case '\n': os << "\\n"; break;
        case '\t': os << "\\t"; break;
        case '\r': os << "\\r"; break;
        default: os << c;
        }
    }
    os << '"';
}

struct Parser {
    const std::string& s;
    size_t i;
    std::string err;

    void ws()
    {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r'))
            i++;
    }
    bool fail(const std::string& what)
    {
        if (err.empty())
            err = what + " at offset " + std::to_string(i);
        return false;
    }
    bool lit(const char* w)
    {
        const size_t n = std::char_traits<char>::length(w);
        if (s.compare(i, n, w) != 0)
            return fail(std::string("expected '") + w + "'");
        i += n;
        return true;
    }
    bool str(std::string& out)
    {
        if (i >= s.size() || s[i] != '"')
            return fail("expected string");
        for (i++; i < s.size() && s[i] != '"'; i++) {
            if (s[i] != '\\') {
                out += s[i];
                continue;
            }
            if (++i >= s.size())
                break;
            switch (s[i]) {
            case 'n': out += '\n'; break;
            case 't': out += '\t'; break;
            case 'r': out += '\r'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'u': {
                if (i + 4 >= s.size())
                    return fail("short \\u escape");
                const unsigned cp = static_cast<unsigned>(std::strtoul(s.substr(i + 1, 4).c_str(), nullptr, 16));
                i += 4;
                if (cp < 0x80)
                    out += static_cast<char>(cp);
                else if (cp < 0x800) {
                    out += static_cast<char>(0xC0 | (cp >> 6));
                    out += static_cast<char>(0x80 | (cp & 0x3F));
                }
                else {
                    out += static_cast<char>(0xE0 | (cp >> 12));
                    out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
                    out += static_cast<char>(0x80 | (cp & 0x3F));
                }
                break;
            }
            default: out += s[i];
            }
        }
        if (i >= s.size())
            return fail("unterminated string");
        i++;
        return true;
    }
    bool val(value& out)
    {
        ws();
        if (i >= s.size())
            return fail("unexpected end of input");
        const char c = s[i];
        if (c == '{') {
            object o;
            i++;
            ws();
            if (i < s.size() && s[i] == '}') {
                i++;
                out = value(o);
                return true;
            }
            for (;;) {
                ws();
                std::string key;
                if (!str(key))
                    return false;
                ws();
                if (i >= s.size() || s[i] != ':')
                    return fail("expected ':'");
                i++;
                value v;
                if (!val(v))
                    return false;
                o[key] = v;
                ws();
                if (i < s.size() && s[i] == ',') {
                    i++;
                    continue;
                }
                if (i < s.size() && s[i] == '}') {
                    i++;
                    break;
                }
                return fail("expected ',' or '}'");
            }
            out = value(o);
            return true;
        }
        if (c == '[') {
            array a;
            i++;
            ws();
            if (i < s.size() && s[i] == ']') {
                i++;
                out = value(a);
                return true;
            }
            for (;;) {
                value v;
                if (!val(v))
                    return false;
                a.push_back(v);
                ws();
                if (i < s.size() && s[i] == ',') {
                    i++;
                    continue;
                }
                if (i < s.size() && s[i] == ']') {
                    i++;
                    break;
                }
                return fail("expected ',' or ']'");
            }
            out = value(a);
            return true;
        }
        if (c == '"') {
            std::string t;
            if (!str(t))
                return false;
            out = value(t);
            return true;
        }
        if (c == 't') {
            out = value(true);
            return lit("true");
        }
        if (c == 'f') {
            out = value(false);
            return lit("false");
        }
        if (c == 'n') {
            out = value();
            return lit("null");
        }
        char* end = nullptr;
        const double d = std::strtod(s.c_str() + i, &end);
        if (end == s.c_str() + i)
            return fail("unexpected character");
        i = static_cast<size_t>(end - s.c_str());
        out = value(d);
        return true;
    }
};
}  // namespace exec_detail

inline std::string value::serialize(bool) const
{
    std::ostringstream os;
    switch (kind_) {
    case NUL: os << "null"; break;
    case BOOL: os << (b_ ? "true" : "false"); break;
    case NUM:
        if (d_ == static_cast<double>(static_cast<long long>(d_)))
            os << static_cast<long long>(d_);
        else
            os << d_;
        break;
    case STR: exec_detail::quote(os, *s_); break;
    case ARR: {
        os << '[';
        bool first = true;
        for (auto&& e : *a_) {
            if (!first)
                os << ',';
            first = false;
            os << e.serialize();
        }
        os << ']';
        break;
    }
    case OBJ: {
        os << '{';
        bool first = true;
        for (auto&& [k, v] : *o_) {
            if (!first)
                os << ',';
            first = false;
            exec_detail::quote(os, k);
            os << ':' << v.serialize();
        }
        os << '}';
        break;
    }
    }
    return os.str();
}

inline std::string parse(value& out, const std::string& text)
{
    exec_detail::Parser p{text, 0, {}};
    if (!p.val(out))
        return p.err.empty() ? std::string("parse error") : p.err;
    return {};
}

inline std::string parse(value& out, std::istream& is)
{
    const std::string text{std::istreambuf_iterator<char>(is), std::istreambuf_iterator<char>()};
    return parse(out, text);
}

inline std::ostream& operator<<(std::ostream& os, const value& v)
{
    return os << v.serialize();
}

inline std::istream& operator>>(std::istream& is, value& v)
{
    const std::string err = parse(v, is);
    if (!err.empty())
        is.setstate(std::ios::failbit);
    return is;
}
}  // namespace picojson
