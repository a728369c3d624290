// toml.hpp — a small TOML reader for the two TOML dialects Iyokan consumes:
//   blueprints      /root/reference/src/iyokan.hpp:1691-1895  ([[file]], [[builtin]], [connect] with quoted keys)
//   plain packets   /root/reference/src/iyokan-packet.cpp:191-233  ([[bits]] / [[ram]] / [[rom]] with byte arrays,
//                   `cycles = N`, and the inline-table form `bits = [ {bytes=[..], size=16, name=".."}, ... ]`
//                   that `iyokan-packet packet2toml` writes)
// Upstream uses toml11; it is not available here and a maintainer keeps toml11 — this is the stand-in that
// lets the C++ frontend run the reference's own fixtures.  Supported: comments, bare / quoted (dotted not
// needed) keys, basic and literal strings, decimal / hex / octal / binary integers with `_`, booleans, floats
// (kept as double), arrays (multi-line, trailing comma, nested), inline tables, [table] and [[array of
// tables]] headers.  Not supported (never used by the fixtures): dates, multi-line strings, dotted keys.
#pragma once
#include <cctype>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "engine.hpp"  // die()

namespace iyk {
namespace host {
namespace toml {

struct Value {
    enum Kind { None, Bool, Int, Float, String, Array, Table } kind = None;
    bool b = false;
    int64_t i = 0;
    double f = 0;
    std::string s;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> tbl;  // insertion order

    const Value* find(const std::string& key) const
    {
        if (kind != Table) return nullptr;
        for (auto& kv : tbl)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    Value* find(const std::string& key) { return const_cast<Value*>(static_cast<const Value*>(this)->find(key)); }
    const Value& at(const std::string& key) const
    {
        const Value* v = find(key);
        if (!v) die("Invalid TOML: missing key \"" + key + "\"");
        return *v;
    }
    int64_t asInt() const
    {
        if (kind != Int) die("Invalid TOML: integer expected");
        return i;
    }
    const std::string& asString() const
    {
        if (kind != String) die("Invalid TOML: string expected");
        return s;
    }
    const std::vector<Value>& asArray() const
    {
        if (kind != Array) die("Invalid TOML: array expected");
        return arr;
    }
    // [[name]] sections, or `name = [ {..}, {..} ]`: both are arrays of tables
    const std::vector<Value>& tables(const std::string& key) const
    {
        static const std::vector<Value> empty;
        const Value* v = find(key);
        if (!v) return empty;
        if (v->kind != Array) die("Invalid TOML: \"" + key + "\" is not an array of tables");
        return v->arr;
    }
};

class Parser {
    const std::string& s_;
    size_t i_ = 0;
    int line_ = 1;

    [[noreturn]] void fail(const std::string& what) { die("Invalid TOML: " + what + " at line " + std::to_string(line_)); }
    bool eof() const { return i_ >= s_.size(); }
    char peek() const { return eof() ? '\0' : s_[i_]; }
    void skipSpace()  // blanks on the current line
    {
        while (!eof() && (s_[i_] == ' ' || s_[i_] == '\t')) ++i_;
    }
    void skipComment()
    {
        if (peek() == '#')
            while (!eof() && s_[i_] != '\n') ++i_;
    }
    void skipBlank()  // blanks, comments and newlines (inside arrays, between statements)
    {
        for (;;) {
            skipSpace();
            skipComment();
            if (peek() == '\n') { ++i_; ++line_; }
            else if (peek() == '\r') ++i_;
            else break;
        }
    }
    void endOfLine()
    {
        skipSpace();
        skipComment();
        if (peek() == '\r') ++i_;
        if (eof()) return;
        if (peek() != '\n') fail("unexpected text after value");
        ++i_;
        ++line_;
    }
    std::string quoted()
    {
        const char q = s_[i_++];
        std::string out;
        while (!eof() && s_[i_] != q) {
            char c = s_[i_++];
            if (c == '\n') fail("newline in string");
            if (q == '"' && c == '\\') {
                if (eof()) fail("bad escape");
                char e = s_[i_++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                default: fail("unsupported escape");
                }
            }
            else
                out += c;
        }
        if (eof()) fail("unterminated string");
        ++i_;
        return out;
    }
    std::string key()
    {
        skipSpace();
        if (peek() == '"' || peek() == '\'') return quoted();
        std::string k;
        while (!eof() && (std::isalnum((unsigned char)s_[i_]) || s_[i_] == '_' || s_[i_] == '-')) k += s_[i_++];
        if (k.empty()) fail("key expected");
        return k;
    }
    Value number()
    {
        std::string tok;
        while (!eof() && (std::isalnum((unsigned char)s_[i_]) || s_[i_] == '_' || s_[i_] == '+' || s_[i_] == '-' || s_[i_] == '.'))
            if (s_[i_] != '_') tok += s_[i_++];
            else ++i_;
        if (tok.empty()) fail("value expected");
        Value v;
        if (tok == "true" || tok == "false") {
            v.kind = Value::Bool;
            v.b = tok == "true";
            return v;
        }
        size_t pos = 0;
        const bool neg = tok[0] == '-';
        const std::string mag = (tok[0] == '-' || tok[0] == '+') ? tok.substr(1) : tok;
        try {
            if (mag.size() > 2 && mag[0] == '0' && (mag[1] == 'x' || mag[1] == 'o' || mag[1] == 'b')) {
                const int base = mag[1] == 'x' ? 16 : mag[1] == 'o' ? 8 : 2;
                v.kind = Value::Int;
                v.i = (int64_t)std::stoull(mag.substr(2), &pos, base);
                if (pos != mag.size() - 2) fail("bad integer " + tok);
                if (neg) v.i = -v.i;
            }
            else if (mag.find_first_of(".eE") != std::string::npos && mag.find_first_not_of("0123456789.eE+-") == std::string::npos) {
                v.kind = Value::Float;
                v.f = std::stod(tok, &pos);
                if (pos != tok.size()) fail("bad float " + tok);
            }
            else {
                v.kind = Value::Int;
                v.i = std::stoll(tok, &pos, 10);
                if (pos != tok.size()) fail("bad integer " + tok);
            }
        }
        catch (const std::exception&) {
            fail("bad number " + tok);
        }
        return v;
    }
    Value value()
    {
        skipSpace();
        Value v;
        const char c = peek();
        if (c == '"' || c == '\'') {
            v.kind = Value::String;
            v.s = quoted();
        }
        else if (c == '[') {
            ++i_;
            v.kind = Value::Array;
            for (;;) {
                skipBlank();
                if (peek() == ']') { ++i_; break; }
                v.arr.push_back(value());
                skipBlank();
                if (peek() == ',') { ++i_; continue; }
                if (peek() == ']') { ++i_; break; }
                fail("',' or ']' expected in array");
            }
        }
        else if (c == '{') {
            ++i_;
            v.kind = Value::Table;
            for (;;) {
                skipSpace();
                if (peek() == '}') { ++i_; break; }
                std::string k = key();
                skipSpace();
                if (peek() != '=') fail("'=' expected in inline table");
                ++i_;
                if (v.find(k)) fail("duplicate key " + k);
                v.tbl.emplace_back(std::move(k), value());
                skipSpace();
                if (peek() == ',') { ++i_; continue; }
                if (peek() == '}') { ++i_; break; }
                fail("',' or '}' expected in inline table");
            }
        }
        else
            v = number();
        return v;
    }

public:
    explicit Parser(const std::string& s) : s_(s) {}
    Value document()
    {
        Value root;
        root.kind = Value::Table;
        Value* cur = &root;
        for (;;) {
            skipBlank();
            if (eof()) break;
            if (peek() == '[') {
                ++i_;
                const bool aot = peek() == '[';
                if (aot) ++i_;
                const std::string name = key();
                skipSpace();
                if (peek() != ']') fail("']' expected");
                ++i_;
                if (aot) {
                    if (peek() != ']') fail("']]' expected");
                    ++i_;
                }
                endOfLine();
                Value* slot = root.find(name);
                if (aot) {
                    if (!slot) {
                        Value a;
                        a.kind = Value::Array;
                        root.tbl.emplace_back(name, std::move(a));
                        slot = &root.tbl.back().second;
                    }
                    if (slot->kind != Value::Array) fail("\"" + name + "\" redefined as an array of tables");
                    Value t;
                    t.kind = Value::Table;
                    slot->arr.push_back(std::move(t));
                    cur = &slot->arr.back();
                }
                else {
                    if (slot) fail("table \"" + name + "\" defined twice");
                    Value t;
                    t.kind = Value::Table;
                    root.tbl.emplace_back(name, std::move(t));
                    cur = &root.tbl.back().second;
                }
                continue;
            }
            std::string k = key();
            skipSpace();
            if (peek() != '=') fail("'=' expected after key \"" + k + "\"");
            ++i_;
            if (cur->find(k)) fail("duplicate key \"" + k + "\"");
            Value v = value();
            endOfLine();
            cur->tbl.emplace_back(std::move(k), std::move(v));
        }
        return root;
    }
};

inline Value parse(const std::string& text)
{
    Parser p(text);
    return p.document();
}

}  // namespace toml
}  // namespace host
}  // namespace iyk
