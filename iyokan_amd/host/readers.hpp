// readers.hpp — netlist readers for the host runtime: Iyokan-L1 JSON and Yosys JSON into any
// NetworkBuilder (plain or HIP), the C++ side of SURVEY.md §8f rank 2.
//
// Same contracts as the reference's readers (behaviour, vocabulary, error texts), written against this
// repository's engine.hpp instead of picojson + the upstream builder:
//   IyokanL1JSONReader::read(builder, istream)   /root/reference/src/iyokan.hpp:2354-2482
//       { "ports": [ {type: "input"|"output", id, portName, portBit, bits: [driver ids]} ],
//         "cells": [ {type: AND|NAND|ANDNOT|OR|NOR|ORNOT|XOR|XNOR|NOT|MUX|DFFP|RAM, id,
//                     input: {A, B, S | D}, ramAddress, ramBit} ] }
//   YosysJSONReader::read(builder, istream)      /root/reference/src/iyokan.hpp:2064-2352
//       one module; ports named `clock` (and an unconnected `reset`) are skipped; cells $_AND_ .. $_MUX_
//       (A, B, S), $_NOT_, $_DFF_P_ (D, Q); constant bits "0" / "1" become CONSTZERO / CONSTONE
//       drivers; $_SDFF_PP0_ / $_SDFF_PP1_ are rejected like upstream ("use $_DFF_P_ instead").
// The JSON parser below is a minimal recursive-descent one (objects, arrays, strings, numbers,
// true / false / null) — enough for the two formats; objects keep insertion order.
#pragma once
#include <cctype>
#include <cstdint>
#include <istream>
#include <iterator>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "engine.hpp"

namespace iyk {
namespace host {
namespace json {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;  // insertion order (Yosys cell order matters for ids only)

    bool isString() const { return kind == String; }
    bool isNumber() const { return kind == Number; }
    const Value* find(const std::string& key) const
    {
        if (kind != Object) return nullptr;
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Value& at(const std::string& key) const
    {
        const Value* v = find(key);
        if (!v) die("Invalid JSON: missing key \"" + key + "\"");
        return *v;
    }
    int asInt() const
    {
        if (kind != Number) die("Invalid JSON: number expected");
        return (int)num;
    }
    const std::string& asString() const
    {
        if (kind != String) die("Invalid JSON: string expected");
        return str;
    }
    const std::vector<Value>& asArray() const
    {
        if (kind != Array) die("Invalid JSON: array expected");
        return arr;
    }
};

class Parser {
    const std::string& s_;
    size_t i_ = 0;

    [[noreturn]] void fail(const char* what) { die(std::string("Invalid JSON: ") + what + " at offset " + std::to_string(i_)); }
    void ws()
    {
        while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_;
    }
    bool eat(char c)
    {
        ws();
        if (i_ < s_.size() && s_[i_] == c) {
            ++i_;
            return true;
        }
        return false;
    }
    std::string string()
    {
        std::string out;
        if (!eat('"')) fail("string expected");
        while (i_ < s_.size() && s_[i_] != '"') {
            char c = s_[i_++];
            if (c == '\\') {
                if (i_ >= s_.size()) fail("bad escape");
                char e = s_[i_++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u':  // netlist names are ASCII: keep the escape verbatim
                    out += "\\u";
                    break;
                default: out += e;
                }
            }
            else
                out += c;
        }
        if (i_ >= s_.size()) fail("unterminated string");
        ++i_;
        return out;
    }

public:
    explicit Parser(const std::string& s) : s_(s) {}
    Value value()
    {
        Value v;
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        const char c = s_[i_];
        if (c == '{') {
            ++i_;
            v.kind = Value::Object;
            if (eat('}')) return v;
            do {
                std::string k = string();
                if (!eat(':')) fail("':' expected");
                v.obj.emplace_back(std::move(k), value());
            } while (eat(','));
            if (!eat('}')) fail("'}' expected");
        }
        else if (c == '[') {
            ++i_;
            v.kind = Value::Array;
            if (eat(']')) return v;
            do {
                v.arr.push_back(value());
            } while (eat(','));
            if (!eat(']')) fail("']' expected");
        }
        else if (c == '"') {
            v.kind = Value::String;
            v.str = string();
        }
        else if (s_.compare(i_, 4, "true") == 0) {
            v.kind = Value::Bool;
            v.b = true;
            i_ += 4;
        }
        else if (s_.compare(i_, 5, "false") == 0) {
            v.kind = Value::Bool;
            i_ += 5;
        }
        else if (s_.compare(i_, 4, "null") == 0) {
            i_ += 4;
        }
        else {
            size_t used = 0;
            try {
                v.num = std::stod(s_.substr(i_, 64), &used);
            }
            catch (...) {
                fail("value expected");
            }
            v.kind = Value::Number;
            i_ += used;
        }
        return v;
    }
    void end()
    {
        ws();
        if (i_ != s_.size()) fail("trailing characters");
    }
};

inline Value parse(std::istream& is)
{
    const std::string text((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    Parser p(text);
    Value v = p.value();
    p.end();
    return v;
}

}  // namespace json

// ------------------------------------------------------------------------------------------
class IyokanL1JSONReader {
public:
    template <class Builder>
    static void read(Builder& b, std::istream& is, int ramWidth = 0)
    {
        const json::Value root = json::parse(is);
        const auto& ports = root.at("ports").asArray();
        const auto& cells = root.at("cells").asArray();
        std::unordered_map<int, int> id2node;

        if (ramWidth == 0) {  // width of the RAM words = 1 + the largest ramBit
            for (auto& c : cells)
                if (c.at("type").asString() == "RAM") ramWidth = std::max(ramWidth, 1 + c.at("ramBit").asInt());
        }
        for (auto& p : ports) {
            const std::string& type = p.at("type").asString();
            const int id = p.at("id").asInt();
            const std::string& name = p.at("portName").asString();
            const int bit = p.at("portBit").asInt();
            if (type == "input")
                id2node[id] = b.INPUT(name, bit);
            else if (type == "output")
                id2node[id] = b.OUTPUT(name, bit);
            else
                die("Invalid JSON of network. Invalid port type: " + type);
        }
        for (auto& c : cells) {
            const std::string& type = c.at("type").asString();
            const int id = c.at("id").asInt();
            int node;
            if (type == "AND") node = b.AND();
            else if (type == "NAND") node = b.NAND();
            else if (type == "ANDNOT") node = b.ANDNOT();
            else if (type == "OR") node = b.OR();
            else if (type == "NOR") node = b.NOR();
            else if (type == "ORNOT") node = b.ORNOT();
            else if (type == "XOR") node = b.XOR();
            else if (type == "XNOR") node = b.XNOR();
            else if (type == "NOT") node = b.NOT();
            else if (type == "MUX") node = b.MUX();
            else if (type == "DFFP") node = b.DFF();
            else if (type == "RAM") node = b.RAM(c.at("ramAddress").asInt(), c.at("ramBit").asInt(), ramWidth);
            else die("Invalid JSON of network. Invalid type: " + type);
            id2node[id] = node;
        }
        auto node = [&](const json::Value& v) {
            auto it = id2node.find(v.asInt());
            if (it == id2node.end()) die("Invalid JSON of network. Unknown id: " + std::to_string(v.asInt()));
            return it->second;
        };
        for (auto& p : ports) {
            if (p.at("type").asString() != "output") continue;
            for (auto& src : p.at("bits").asArray()) b.connect(node(src), node(p.at("id")));
        }
        for (auto& c : cells) {
            const std::string& type = c.at("type").asString();
            const int dst = node(c.at("id"));
            const json::Value& in = c.at("input");
            if (type == "DFFP" || type == "RAM")
                b.connect(node(in.at("D")), dst);
            else if (type == "NOT")
                b.connect(node(in.at("A")), dst);
            else if (type == "MUX") {
                b.connect(node(in.at("A")), dst);
                b.connect(node(in.at("B")), dst);
                b.connect(node(in.at("S")), dst);
            }
            else {
                b.connect(node(in.at("A")), dst);
                b.connect(node(in.at("B")), dst);
            }
        }
    }
};

class YosysJSONReader {
public:
    template <class Builder>
    static void read(Builder& b, std::istream& is)
    {
        const json::Value root = json::parse(is);
        const json::Value& modules = root.at("modules");
        if (modules.kind != json::Value::Object || modules.obj.size() != 1)
            die(".modules should be an object of size 1");
        const json::Value& mod = modules.obj[0].second;

        std::unordered_map<int, int> bit2node;  // yosys net number -> driver node
        int constNode[2] = {-1, -1};
        auto constant = [&](const std::string& v) {
            const int k = (v == "1") ? 1 : 0;
            if (v != "0" && v != "1") die("Invalid constant bit in Yosys JSON: " + v);
            if (constNode[k] < 0) constNode[k] = k ? b.CONSTONE() : b.CONSTZERO();
            return constNode[k];
        };
        struct Pending {
            int node;
            const json::Value* src;
        };
        std::vector<Pending> edges;  // (destination node, yosys bit) pairs, connected once every driver is known

        for (auto& kv : mod.at("ports").obj) {
            const std::string& name = kv.first;
            const std::string& dir = kv.second.at("direction").asString();
            const auto& bits = kv.second.at("bits").asArray();
            if (name == "clock" || (name == "reset" && bits.empty())) continue;
            for (size_t i = 0; i < bits.size(); ++i) {
                if (dir == "input") {
                    if (bits[i].isString()) die("input port bit tied to a constant");
                    bit2node[bits[i].asInt()] = b.INPUT(name, (int)i);
                }
                else if (dir == "output")
                    edges.push_back({b.OUTPUT(name, (int)i), &bits[i]});
                else
                    die("Invalid direction token: " + dir);
            }
        }
        for (auto& kv : mod.at("cells").obj) {
            const json::Value& cell = kv.second;
            const std::string& type = cell.at("type").asString();
            const json::Value& conn = cell.at("connections");
            auto pin = [&](const char* p) -> const json::Value* { return &conn.at(p).asArray().at(0); };
            int node;
            bool dff = false, mux = false, unary = false;
            if (type == "$_AND_") node = b.AND();
            else if (type == "$_NAND_") node = b.NAND();
            else if (type == "$_ANDNOT_") node = b.ANDNOT();
            else if (type == "$_OR_") node = b.OR();
            else if (type == "$_NOR_") node = b.NOR();
            else if (type == "$_ORNOT_") node = b.ORNOT();
            else if (type == "$_XOR_") node = b.XOR();
            else if (type == "$_XNOR_") node = b.XNOR();
            else if (type == "$_NOT_") { node = b.NOT(); unary = true; }
            else if (type == "$_MUX_") { node = b.MUX(); mux = true; }
            else if (type == "$_DFF_P_") { node = b.DFF(); dff = true; }
            else if (type == "$_SDFF_PP0_" || type == "$_SDFF_PP1_")
                die(type + " is not supported. Please use $_DFF_P_ instead (dfflegalize -cell $_DFF_P_ 01 in Yosys).");
            else die("Invalid JSON of network. Invalid cell type: " + type + " (" + kv.first + ")");
            const json::Value* out = pin(dff ? "Q" : "Y");
            if (out->isString()) die("cell output tied to a constant: " + kv.first);
            bit2node[out->asInt()] = node;
            if (dff)
                edges.push_back({node, pin("D")});
            else {
                edges.push_back({node, pin("A")});
                if (!unary) edges.push_back({node, pin("B")});
                if (mux) edges.push_back({node, pin("S")});
            }
        }
        for (auto& e : edges) {
            int src;
            if (e.src->isString())
                src = constant(e.src->asString());
            else {
                auto it = bit2node.find(e.src->asInt());
                if (it == bit2node.end()) die("Invalid JSON of network. Undriven bit: " + std::to_string(e.src->asInt()));
                src = it->second;
            }
            b.connect(src, e.node);
        }
    }
};

// readNetworkFromJSON of the reference's tests (/root/reference/src/test0.cpp:21-27): an L1 netlist
// straight into a network of the builder's backend.
template <class Builder, class Factory>
typename Builder::NetworkType readNetworkFromJSON(Factory& f, std::istream& is)
{
    Builder b(f);
    IyokanL1JSONReader::read(b, is);
    return b.build();
}
template <class Builder, class Factory>
typename Builder::NetworkType readNetworkFromYosysJSON(Factory& f, std::istream& is)
{
    Builder b(f);
    YosysJSONReader::read(b, is);
    return b.build();
}

}  // namespace host
}  // namespace iyk
