// packet.hpp — Iyokan's request / result packets on the host side of the HIP backend.
//
//   PlainPacket   named bit vectors `bits`, RAM / ROM images, optional cycle count
//                 (/root/reference/src/packet.hpp:193-206)
//   TFHEPacket    the same, encrypted: bits / ramInTLWE / romInTLWE as TLWE lvl0 vectors, ram / rom as TRLWE lvl1
//                 vectors for the CMUX memories (/root/reference/src/packet.hpp:208-223)
// and their two wire formats:
//   TOML          `iyokan-packet toml2packet / packet2toml` (/root/reference/src/iyokan-packet.cpp:28-142,191-233)
//   binary        cereal::PortableBinary{Output,Input}Archive, readFromArchive / writeToArchive
//                 (/root/reference/src/packet.hpp:287-344)
//
// cereal is not available in this container, so the binary reader / writer below restate its PortableBinary
// encoding for exactly the types these packets contain [recollection of cereal 1.3's published rules; the
// reference tree holds no binary fixture to pin them against — tests round-trip self-written archives and
// hand-assembled byte strings]:
//   archive            1 byte: 1 = payload is little-endian (the writer's default), then the object
//   struct             its serialize() arguments, in order, no framing
//   bool, Bit          1 byte (enum class Bit : bool is saved as its underlying type)
//   int32 / uint32 / uint64   raw little-endian bytes
//   size tag           uint64
//   std::string        size tag + bytes
//   std::vector<T>     size tag + elements (one block of raw bytes when T is arithmetic)
//   std::array<T, N>   N elements, no size tag (TLWE lvl0 = array<u32, n+1>; TRLWE lvl1 = array<array<u32, N>, 2>)
//   std::unordered_map<K, V>   size tag + (key, value) pairs in iteration order
//   std::optional<T>   1 byte `nullopt` flag (1 = empty), then T when present
// The TLWE / TRLWE array lengths are not in the stream: they come from the parameter set (compile-time in
// upstream, iyk_params here), which is why every reader takes the params.
//
// Key material: TFHEpp serialises SecretKey / EvalKey with its own serialize() members, whose layout this
// repository cannot pin; upstream code loads them through TFHEpp and hands the raw arrays
// (`ek.getbk<lvl01param>()`, `ek.getiksk<lvl10param>()`) to iyk_hip_init.  For this
// repository's own tools the same container rules carry a plain `KeyArchive` (params + s0 + s1 + bk + ksk).
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include <fstream>
#include <istream>
#include <map>
#include <optional>
#include <ostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/iyokan_hip_params.h"
#include "engine.hpp"
#include "toml.hpp"

namespace iyk {
namespace host {

using Bit = uint8_t;  // 0 / 1
using TLWEVec = std::vector<uint32_t>;  // count * (n+1) words, row-major

struct PlainPacket {
    std::map<std::string, std::vector<Bit>> ram, rom, bits;
    std::optional<int> numCycles;
    bool operator==(const PlainPacket& o) const { return ram == o.ram && rom == o.rom && bits == o.bits && numCycles == o.numCycles; }
};

struct TFHEPacket {
    std::map<std::string, std::vector<uint32_t>> ram, rom;              // TRLWE lvl1: count * 2N words
    std::map<std::string, TLWEVec> ramInTLWE, romInTLWE, bits;         // TLWE lvl0: count * (n+1) words
    std::optional<int> numCycles;
    bool operator==(const TFHEPacket& o) const
    {
        return ram == o.ram && rom == o.rom && ramInTLWE == o.ramInTLWE && romInTLWE == o.romInTLWE && bits == o.bits &&
               numCycles == o.numCycles;
    }
};

struct KeyArchive {  // this repository's own key container (not TFHEpp's)
    iyk_params params{};
    std::vector<uint32_t> s0, s1, bk, ksk;  // s0 / s1 empty in an evaluation-key archive
};

// ---- TOML form --------------------------------------------------------------------------------------------
// doToml2Packet (/root/reference/src/iyokan-packet.cpp:191-233): `size` bits, filled from `bytes` lsb first;
// missing bytes read as 0, surplus bytes are ignored, only the low 8 bits of a byte count.
inline std::vector<Bit> bitsFromBytes(const std::vector<toml::Value>& bytes, size_t size)
{
    std::vector<Bit> out(size, 0);
    for (size_t i = 0; i < size; ++i) {
        const size_t byte = i / 8;
        if (byte >= bytes.size()) break;
        out[i] = (Bit)(((uint64_t)bytes[byte].asInt() & 0xFF) >> (i % 8) & 1u);
    }
    return out;
}
inline std::vector<uint8_t> bytesFromBits(const std::vector<Bit>& bits)
{
    std::vector<uint8_t> out((bits.size() + 7) / 8, 0);
    for (size_t i = 0; i < bits.size(); ++i) out[i / 8] |= (uint8_t)((bits[i] & 1u) << (i % 8));
    return out;
}

inline PlainPacket plainPacketFromTOML(const std::string& text)
{
    const toml::Value root = toml::parse(text);
    PlainPacket pkt;
    if (const toml::Value* c = root.find("cycles")) {
        if (c->asInt() >= 0) pkt.numCycles = (int)c->asInt();
    }
    auto load = [&](const char* kind, std::map<std::string, std::vector<Bit>>& dst) {
        for (const toml::Value& e : root.tables(kind)) {
            const std::string& name = e.at("name").asString();
            const int64_t size = e.at("size").asInt();
            if (size < 0) die(std::string("Invalid packet: negative size of ") + kind + " " + name);
            if (!dst.emplace(name, bitsFromBytes(e.at("bytes").asArray(), (size_t)size)).second)
                die(std::string("Invalid packet: duplicate ") + kind + " entry " + name);
        }
    };
    load("ram", pkt.ram);
    load("rom", pkt.rom);
    load("bits", pkt.bits);
    return pkt;
}
inline std::string readTextFile(const std::string& path)
{
    std::ifstream ifs(path, std::ios::binary);
    if (!ifs) die("Can't open the file to read from; Maybe not found?: " + path);
    std::ostringstream ss;
    ss << ifs.rdbuf();
    return ss.str();
}
inline PlainPacket plainPacketFromTOMLFile(const std::string& path) { return plainPacketFromTOML(readTextFile(path)); }

// doPacket2Toml shape (/root/reference/src/iyokan-packet.cpp:96-142)
inline std::string plainPacketToTOML(const PlainPacket& pkt)
{
    std::ostringstream os;
    if (pkt.numCycles) os << "cycles = " << *pkt.numCycles << "\n";
    auto dump = [&](const char* kind, const std::map<std::string, std::vector<Bit>>& src) {
        for (auto& kv : src) {
            os << "[[" << kind << "]]\nname = \"" << kv.first << "\"\nsize = " << kv.second.size() << "\nbytes = [";
            const auto bytes = bytesFromBits(kv.second);
            for (size_t i = 0; i < bytes.size(); ++i) os << (i ? ", " : "") << (int)bytes[i];
            os << "]\n";
        }
    };
    dump("rom", pkt.rom);
    dump("ram", pkt.ram);
    dump("bits", pkt.bits);
    return os.str();
}

// toml2packet(got) == toml2packet(expected) of the reference's test driver (/root/reference/test.rb:34-68):
// the same named entries with the same (size, zero-padded bytes), and the same cycle count
inline bool samePacketContent(const PlainPacket& a, const PlainPacket& b, std::string* why = nullptr)
{
    auto cmp = [&](const char* kind, const std::map<std::string, std::vector<Bit>>& x, const std::map<std::string, std::vector<Bit>>& y) {
        for (auto& kv : x) {
            auto it = y.find(kv.first);
            if (it == y.end() || it->second != kv.second) {
                if (why) *why += std::string(kind) + "." + kv.first + " differs; ";
                return false;
            }
        }
        if (x.size() != y.size()) {
            if (why) *why += std::string(kind) + ": different entry sets; ";
            return false;
        }
        return true;
    };
    bool ok = cmp("bits", a.bits, b.bits);
    ok = cmp("ram", a.ram, b.ram) && ok;
    ok = cmp("rom", a.rom, b.rom) && ok;
    if (a.numCycles.value_or(-1) != b.numCycles.value_or(-1)) {
        if (why) *why += "cycles differ; ";
        ok = false;
    }
    return ok;
}

// ---- cereal PortableBinary form ---------------------------------------------------------------------------
namespace cereal_pb {

class Writer {
    std::ostream& os_;

public:
    explicit Writer(std::ostream& os) : os_(os) { u8(1); }  // payload is little-endian
    void raw(const void* p, size_t n) { os_.write(static_cast<const char*>(p), (std::streamsize)n); }
    void u8(uint8_t v) { raw(&v, 1); }
    void u32(uint32_t v) { raw(&v, 4); }  // host is little-endian (x86-64): no swap
    void i32(int32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void f64(double v) { raw(&v, 8); }
    void str(const std::string& s)
    {
        u64(s.size());
        raw(s.data(), s.size());
    }
    void optInt(const std::optional<int>& o)
    {
        u8(o ? 0 : 1);  // "nullopt" flag
        if (o) i32(*o);
    }
    void bitMap(const std::map<std::string, std::vector<Bit>>& m)
    {
        u64(m.size());
        for (auto& kv : m) {
            str(kv.first);
            u64(kv.second.size());
            for (Bit b : kv.second) u8(b ? 1 : 0);
        }
    }
    // vector of fixed-size word arrays (TLWE / TRLWE): element count, then the raw words
    void arrayMap(const std::map<std::string, std::vector<uint32_t>>& m, size_t words)
    {
        u64(m.size());
        for (auto& kv : m) {
            if (kv.second.size() % words) die("packet entry " + kv.first + " is not a whole number of ciphertexts");
            str(kv.first);
            u64(kv.second.size() / words);
            raw(kv.second.data(), kv.second.size() * sizeof(uint32_t));
        }
    }
    void u32vec(const std::vector<uint32_t>& v)
    {
        u64(v.size());
        raw(v.data(), v.size() * sizeof(uint32_t));
    }
};

class Reader {
    std::istream& is_;
    bool swap_ = false;  // archive written big-endian
    uint64_t left_ = ~0ull;  // bytes between the read position and the end of the stream (~0: the stream cannot seek)

    [[noreturn]] void bad(const char* what) { die(std::string("Invalid archive: ") + what); }
    // A size tag promises `bytes` more bytes of payload: refuse it BEFORE allocating when the stream does not hold them
    // (a truncated or hostile packet used to reach std::vector with a 4 GiB request and end in bad_alloc / OOM).
    void need(uint64_t bytes)
    {
        if (bytes > left_) bad("size tag exceeds the remaining bytes");
    }

public:
    explicit Reader(std::istream& is) : is_(is)
    {
        const std::istream::pos_type here = is_.tellg();
        if (here != std::istream::pos_type(-1) && is_.seekg(0, std::ios::end)) {
            const std::istream::pos_type end = is_.tellg();
            is_.seekg(here);
            if (end != std::istream::pos_type(-1) && end >= here) left_ = (uint64_t)(end - here);
        }
        is_.clear();
        const uint8_t little = u8();
        if (little > 1) bad("bad endianness flag");
        swap_ = little == 0;
    }
    void raw(void* p, size_t n)
    {
        if (n > left_) bad("truncated");
        is_.read(static_cast<char*>(p), (std::streamsize)n);
        if ((size_t)is_.gcount() != n) bad("truncated");
        if (left_ != ~0ull) left_ -= n;
    }
    template <class T>
    T scalar()
    {
        unsigned char b[sizeof(T)];
        raw(b, sizeof(T));
        if (swap_)
            for (size_t i = 0; i < sizeof(T) / 2; ++i) std::swap(b[i], b[sizeof(T) - 1 - i]);
        T v;
        std::memcpy(&v, b, sizeof(T));
        return v;
    }
    uint8_t u8()
    {
        uint8_t v;
        raw(&v, 1);
        return v;
    }
    uint32_t u32() { return scalar<uint32_t>(); }
    int32_t i32() { return scalar<int32_t>(); }
    uint64_t u64() { return scalar<uint64_t>(); }
    double f64() { return scalar<double>(); }
    uint64_t size(uint64_t limit)
    {
        const uint64_t n = u64();
        if (n > limit) bad("implausible size tag");
        return n;
    }
    std::string str()
    {
        const uint64_t len = size(1u << 20);
        need(len);
        std::string s((size_t)len, '\0');
        if (!s.empty()) raw(&s[0], s.size());
        return s;
    }
    std::optional<int> optInt()
    {
        const uint8_t nullopt = u8();
        if (nullopt > 1) bad("bad optional flag");
        if (nullopt) return std::nullopt;
        return (int)i32();
    }
    void words(std::vector<uint32_t>& v, size_t n)
    {
        if (n > (~0ull) / sizeof(uint32_t)) bad("implausible size tag");
        need((uint64_t)n * sizeof(uint32_t));
        v.resize(n);
        if (n) raw(v.data(), n * sizeof(uint32_t));
        if (swap_)
            for (auto& w : v) w = __builtin_bswap32(w);
    }
    void bitMap(std::map<std::string, std::vector<Bit>>& m)
    {
        const uint64_t n = size(1u << 20);
        need(n * 16);  // every entry carries at least two 8-byte size tags
        for (uint64_t e = 0; e < n; ++e) {
            std::string key = str();
            const uint64_t nbits = size(1ull << 32);
            need(nbits);
            std::vector<Bit> v((size_t)nbits);
            if (!v.empty()) raw(v.data(), v.size());
            for (Bit& b : v)
                if (b > 1) bad("Bit value out of range");
            if (!m.emplace(std::move(key), std::move(v)).second) bad("duplicate key");
        }
    }
    void arrayMap(std::map<std::string, std::vector<uint32_t>>& m, size_t wordsPer)
    {
        const uint64_t n = size(1u << 20);
        need(n * 16);
        for (uint64_t e = 0; e < n; ++e) {
            std::string key = str();
            const uint64_t count = size((1ull << 34) / wordsPer);
            std::vector<uint32_t> v;
            words(v, (size_t)count * wordsPer);
            if (!m.emplace(std::move(key), std::move(v)).second) bad("duplicate key");
        }
    }
    void u32vec(std::vector<uint32_t>& v, uint64_t limit) { words(v, (size_t)size(limit)); }
    void expectEnd()
    {
        char c;
        is_.read(&c, 1);
        if (is_.gcount() != 0 || (left_ != ~0ull && left_ != 0)) bad("trailing bytes");
    }
};

}  // namespace cereal_pb

// writeToArchive / readFromArchive for the packet types (/root/reference/src/packet.hpp:287-344)
inline void writeToArchive(std::ostream& os, const PlainPacket& p)
{
    cereal_pb::Writer w(os);
    w.bitMap(p.ram);  // serialize(): ar(ram, rom, bits, numCycles)
    w.bitMap(p.rom);
    w.bitMap(p.bits);
    w.optInt(p.numCycles);
}
inline void readFromArchive(PlainPacket& p, std::istream& is)
{
    cereal_pb::Reader r(is);
    p = PlainPacket{};
    r.bitMap(p.ram);
    r.bitMap(p.rom);
    r.bitMap(p.bits);
    p.numCycles = r.optInt();
    r.expectEnd();
}
inline void writeToArchive(std::ostream& os, const TFHEPacket& p, const iyk_params& prm)
{
    cereal_pb::Writer w(os);
    const size_t trlwe = (size_t)(prm.k + 1) * prm.N, tlwe = (size_t)prm.n + 1;
    w.arrayMap(p.ram, trlwe);  // serialize(): ar(ram, ramInTLWE, rom, romInTLWE, bits, numCycles)
    w.arrayMap(p.ramInTLWE, tlwe);
    w.arrayMap(p.rom, trlwe);
    w.arrayMap(p.romInTLWE, tlwe);
    w.arrayMap(p.bits, tlwe);
    w.optInt(p.numCycles);
}
inline void readFromArchive(TFHEPacket& p, std::istream& is, const iyk_params& prm)
{
    cereal_pb::Reader r(is);
    const size_t trlwe = (size_t)(prm.k + 1) * prm.N, tlwe = (size_t)prm.n + 1;
    p = TFHEPacket{};
    r.arrayMap(p.ram, trlwe);
    r.arrayMap(p.ramInTLWE, tlwe);
    r.arrayMap(p.rom, trlwe);
    r.arrayMap(p.romInTLWE, tlwe);
    r.arrayMap(p.bits, tlwe);
    p.numCycles = r.optInt();
    r.expectEnd();
}
inline void writeToArchive(std::ostream& os, const KeyArchive& k)
{
    cereal_pb::Writer w(os);
    const iyk_params& p = k.params;
    for (uint32_t v : {p.n, p.N, p.k, p.l, p.Bgbit, p.t, p.basebit, p.mu}) w.u32(v);
    w.f64(p.alpha0);
    w.f64(p.alpha1);
    w.u32vec(k.s0);
    w.u32vec(k.s1);
    w.u32vec(k.bk);
    w.u32vec(k.ksk);
}
inline void readFromArchive(KeyArchive& k, std::istream& is)
{
    cereal_pb::Reader r(is);
    iyk_params& p = k.params;
    for (uint32_t* v : {&p.n, &p.N, &p.k, &p.l, &p.Bgbit, &p.t, &p.basebit, &p.mu}) *v = r.u32();
    p.alpha0 = r.f64();
    p.alpha1 = r.f64();
    if (p.n == 0 || p.n > 4096 || p.N == 0 || p.N > 65536 || p.k == 0 || p.k > 4 || p.l == 0 || p.l > 16 || p.t == 0 ||
        p.t > 32 || p.basebit == 0 || p.basebit > 8)
        die("Invalid archive: implausible parameter set");
    r.u32vec(k.s0, p.n);
    r.u32vec(k.s1, (uint64_t)p.k * p.N);
    r.u32vec(k.bk, iyk_bk_words(&p));
    r.u32vec(k.ksk, iyk_ksk_words(&p));
    if ((!k.s0.empty() && k.s0.size() != p.n) || (!k.bk.empty() && k.bk.size() != iyk_bk_words(&p)) ||
        (!k.ksk.empty() && k.ksk.size() != iyk_ksk_words(&p)))
        die("Invalid archive: key sizes do not match the parameter set");
    r.expectEnd();
}

// ---- import of TFHEpp's OWN key archives (the files stock `iyokan-packet genkey / genevalkey` write) -----------------
// Twin of iyokan_amd/tfhepp_keys.py — read its header for what is fixed by cereal and by the reference's call sites
// (/root/reference/src/iyokan_cufhe.cpp:546-549, /root/reference/src/iyokan-packet.cpp:150-160), what is ASSUMED (binary
// keys, one uint32 per bit; key.lvl0 then key.lvl1 right behind the endianness byte) and why the EvalKey reader SEARCHES
// for the two blobs instead of walking a struct whose member order cannot be read here.  UNVERIFIED against real TFHEpp.
namespace tfhepp_import {
inline uint32_t u32le(const std::vector<unsigned char>& d, size_t o)
{
    return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24);
}
inline bool plausibleNext(const std::vector<unsigned char>& d, size_t o)
{
    if (o == d.size()) return true;
    if (o + 4 > d.size()) return false;
    const uint32_t w = u32le(d, o);
    if (w == 0 || ((w & 0x80000000u) && (w & 0x7FFFFFFFu) <= 64)) return true;
    if (o + 8 <= d.size()) return w <= 4096 && u32le(d, o + 4) == 0;   // u64 size tag of a small unordered_map
    return false;
}
// offset of the unique `nbytes` blob that sits behind an "object follows" pointer id and before a plausible cereal item;
// 0 = none or ambiguous
inline size_t findBlob(const std::vector<unsigned char>& d, size_t nbytes)
{
    size_t found = 0, count = 0;
    for (size_t h = 0; h + 4 + nbytes <= d.size(); ++h) {
        if (d[h + 3] != 0x80 || d[h + 2] != 0 || d[h + 1] != 0 || d[h] < 1 || d[h] > 64) continue;
        if (!plausibleNext(d, h + 4 + nbytes)) continue;
        found = h + 4;
        ++count;
    }
    return count == 1 ? found : 0;
}
inline std::vector<unsigned char> slurp(const std::string& path)
{
    std::ifstream ifs(path, std::ios::binary);
    if (!ifs) die("Can't open the file to read from; Maybe not found?: " + path);
    ifs.seekg(0, std::ios::end);
    const std::streamoff n = ifs.tellg();
    ifs.seekg(0);
    std::vector<unsigned char> d((size_t)(n < 0 ? 0 : n));
    if (!d.empty()) ifs.read(reinterpret_cast<char*>(d.data()), (std::streamsize)d.size());
    if ((size_t)ifs.gcount() != d.size()) die("Invalid archive: short read: " + path);
    return d;
}
inline void words(const std::vector<unsigned char>& d, size_t o, size_t n, std::vector<uint32_t>& out)
{
    out.resize(n);
    for (size_t i = 0; i < n; ++i) out[i] = u32le(d, o + 4 * i);
}
}  // namespace tfhepp_import

// bk<lvl01param> + iksk<lvl10param> of a TFHEpp::EvalKey archive for parameter set `p`; false when this archive does not
// hold exactly one blob of each size (another parameter set, a truncated file, not an EvalKey)
inline bool readTFHEppEvalKey(const std::vector<unsigned char>& d, const iyk_params& p, KeyArchive& out)
{
    if (d.empty() || d[0] != 1) return false;   // little-endian archives only
    const size_t bkw = iyk_bk_words(&p), kw = iyk_ksk_words(&p);
    const size_t ob = tfhepp_import::findBlob(d, 4 * bkw), ok = tfhepp_import::findBlob(d, 4 * kw);
    if (!ob || !ok) return false;
    if (!(ob + 4 * bkw + 4 <= ok || ok + 4 * kw + 4 <= ob)) return false;
    out = KeyArchive{};
    out.params = p;
    tfhepp_import::words(d, ob, bkw, out.bk);
    tfhepp_import::words(d, ok, kw, out.ksk);
    return true;
}
// key.lvl0 + key.lvl1 of a TFHEpp::SecretKey archive: n + N binary words right behind the endianness byte
inline bool readTFHEppSecretKey(const std::vector<unsigned char>& d, const iyk_params& p, KeyArchive& out)
{
    const size_t need = 1 + 4 * ((size_t)p.n + (size_t)p.k * p.N);
    if (d.size() < need || d[0] != 1) return false;
    KeyArchive k;
    k.params = p;
    tfhepp_import::words(d, 1, p.n, k.s0);
    tfhepp_import::words(d, 1 + 4 * (size_t)p.n, (size_t)p.k * p.N, k.s1);
    for (uint32_t v : k.s0)
        if (v > 1) return false;
    for (uint32_t v : k.s1)
        if (v > 1) return false;
    out = k;
    return true;
}
// sample rows of both keys must decrypt under the secret key (message + noise within 6 sigma): the one check that does
// not depend on any recollection of TFHEpp's layout
inline bool verifyImportedKeys(const KeyArchive& sk, const KeyArchive& ek)
{
    const iyk_params& p = ek.params;
    const size_t n1 = (size_t)p.n + 1, nb = (1u << p.basebit) - 1;
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto next = [&](uint64_t m) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        return (state >> 33) % m;
    };
    auto centred = [](uint32_t d) { return (double)(int32_t)d / 4294967296.0; };
    for (int s = 0; s < 64; ++s) {
        const size_t i = next(p.N), j = next(p.t), v = next(nb);
        const uint32_t* row = ek.ksk.data() + ((i * p.t + j) * nb + v) * n1;
        uint32_t ph = row[p.n];
        for (uint32_t x = 0; x < p.n; ++x) ph -= row[x] * sk.s0[x];
        const uint32_t msg = (sk.s1[i] * (uint32_t)(v + 1)) << (32 - (j + 1) * p.basebit);
        if (std::fabs(centred(ph - msg)) > 6 * p.alpha0 + 1e-9) return false;
    }
    const size_t rows = (size_t)(p.k + 1) * p.l;
    for (int s = 0; s < 64; ++s) {
        const size_t i = next(p.n), r = next(rows);
        const uint32_t* a = ek.bk.data() + ((i * rows + r) * 2) * p.N;
        const uint32_t* b = a + p.N;
        uint32_t as0 = a[0] * sk.s1[0];
        for (uint32_t x = 1; x < p.N; ++x) as0 -= a[x] * sk.s1[p.N - x];   // (a * s1)[0] mod X^N + 1
        const size_t c = r / p.l, j = r % p.l;
        const uint32_t m = sk.s0[i] << (32 - (j + 1) * p.Bgbit);
        const uint32_t want = c == 1 ? m : (0u - m * sk.s1[0]);
        if (std::fabs(centred(b[0] - as0 - want)) > 6 * p.alpha1 + 1e-9) return false;
    }
    return true;
}

template <class T, class... A>
void writeToArchiveFile(const std::string& path, const T& src, const A&... a)
{
    std::ofstream ofs(path, std::ios::binary);
    if (!ofs) die("Unable to write into archive: " + path);
    writeToArchive(ofs, src, a...);
}
template <class T, class... A>
T readFromArchiveFile(const std::string& path, const A&... a)
{
    std::ifstream ifs(path, std::ios::binary);
    if (!ifs) die("Can't open the file to read from; Maybe not found?: " + path);
    T ret;
    readFromArchive(ret, ifs, a...);
    return ret;
}

// ---- encrypt / decrypt of packets (PlainPacket::encrypt / TFHEPacket::decrypt, /root/reference/src/packet.hpp:225-290);
// needs libiyokan_client.so (keygen / enc / dec stand-in for TFHEpp).  Both forms of every memory image, as upstream: TLWE lvl0
// rows (ramInTLWE / romInTLWE: what the MUX memories and this backend read) and TRLWE lvl1 (ram: one ciphertext per bit with
// +-mu in coefficient 0; rom: N bits per ciphertext — what upstream's CMUX memories read), so that a request made here can
// drive either kind of run and a result made upstream decrypts here ------
extern "C" {
int iyk_client_encrypt_bits(const iyk_params*, const uint32_t*, uint64_t, int, const uint8_t*, uint64_t, uint32_t*);
int iyk_client_decrypt_bits(const iyk_params*, const uint32_t*, const uint32_t*, uint64_t, uint8_t*);
int iyk_client_encrypt_trlwe(const iyk_params*, const uint32_t*, uint64_t, int, const uint32_t*, uint64_t, uint32_t*);
int iyk_client_trlwe_phases(const iyk_params*, const uint32_t*, const uint32_t*, uint64_t, uint32_t*);
}

inline TLWEVec encryptBits(const iyk_params& p, const std::vector<uint32_t>& s0, const std::vector<Bit>& src, uint64_t seed = 0,
                           int deterministic = 0)
{
    TLWEVec out(src.size() * ((size_t)p.n + 1));
    if (!src.empty()) iyk_client_encrypt_bits(&p, s0.data(), seed, deterministic, src.data(), src.size(), out.data());
    return out;
}
inline std::vector<Bit> decryptBits(const iyk_params& p, const std::vector<uint32_t>& s0, const TLWEVec& src)
{
    std::vector<Bit> out(src.size() / ((size_t)p.n + 1));
    if (!out.empty()) iyk_client_decrypt_bits(&p, s0.data(), src.data(), out.size(), out.data());
    return out;
}
// encryptRAM / encryptROM (:78-118): `perCt` bits per TRLWE (1: RAM, coefficient 0 only; N: ROM), +-mu, zero padding
inline std::vector<uint32_t> encryptTRLWEBits(const iyk_params& p, const std::vector<uint32_t>& s1, const std::vector<Bit>& src,
                                              size_t perCt, uint64_t seed, int deterministic)
{
    const size_t N = p.N, count = (src.size() + perCt - 1) / perCt;
    std::vector<uint32_t> msg(count * N, 0u), out(count * 2 * N);
    for (size_t i = 0; i < src.size(); ++i) msg[(i / perCt) * N + i % perCt] = src[i] ? p.mu : 0u - p.mu;
    if (count) iyk_client_encrypt_trlwe(&p, s1.data(), seed, deterministic, msg.data(), count, out.data());
    return out;
}
// decryptRAM / decryptROM (:153-183): the sign of coefficient 0 (RAM) or of every coefficient (ROM: a multiple of N bits)
inline std::vector<Bit> decryptTRLWEBits(const iyk_params& p, const std::vector<uint32_t>& s1, const std::vector<uint32_t>& ct,
                                         size_t perCt)
{
    const size_t N = p.N, count = ct.size() / (2 * N);
    std::vector<uint32_t> ph(count * N);
    if (count) iyk_client_trlwe_phases(&p, s1.data(), ct.data(), count, ph.data());
    std::vector<Bit> out;
    out.reserve(count * perCt);
    for (size_t g = 0; g < count; ++g)
        for (size_t i = 0; i < perCt; ++i) out.push_back((int32_t)ph[g * N + i] > 0 ? 1 : 0);
    return out;
}
// s1 empty: TLWE forms only (the callers that only ever feed this backend; 8 KB per RAM bit saved)
inline TFHEPacket encryptPacket(const iyk_params& p, const std::vector<uint32_t>& s0, const PlainPacket& plain, uint64_t seed = 0,
                                int deterministic = 0, const std::vector<uint32_t>& s1 = {})
{
    TFHEPacket t;
    t.numCycles = plain.numCycles;
    uint64_t k = 0;
    for (auto& kv : plain.ram) {
        t.ramInTLWE.emplace(kv.first, encryptBits(p, s0, kv.second, seed + (++k), deterministic));
        if (!s1.empty()) t.ram.emplace(kv.first, encryptTRLWEBits(p, s1, kv.second, 1, seed + (++k), deterministic));
    }
    for (auto& kv : plain.rom) {
        t.romInTLWE.emplace(kv.first, encryptBits(p, s0, kv.second, seed + (++k), deterministic));
        if (!s1.empty()) t.rom.emplace(kv.first, encryptTRLWEBits(p, s1, kv.second, p.N, seed + (++k), deterministic));
    }
    for (auto& kv : plain.bits) t.bits.emplace(kv.first, encryptBits(p, s0, kv.second, seed + (++k), deterministic));
    return t;
}
// TFHEPacket::decrypt: the TRLWE maps first (needs s1), the TLWE maps fill in the names the TRLWE maps do not hold
inline PlainPacket decryptPacket(const iyk_params& p, const std::vector<uint32_t>& s0, const TFHEPacket& t,
                                 const std::vector<uint32_t>& s1 = {})
{
    PlainPacket plain;
    plain.numCycles = t.numCycles;
    if (!s1.empty()) {
        for (auto& kv : t.ram) plain.ram.emplace(kv.first, decryptTRLWEBits(p, s1, kv.second, 1));
        for (auto& kv : t.rom) plain.rom.emplace(kv.first, decryptTRLWEBits(p, s1, kv.second, p.N));
    }
    for (auto& kv : t.ramInTLWE) plain.ram.emplace(kv.first, decryptBits(p, s0, kv.second));
    for (auto& kv : t.romInTLWE) plain.rom.emplace(kv.first, decryptBits(p, s0, kv.second));
    for (auto& kv : t.bits) plain.bits.emplace(kv.first, decryptBits(p, s0, kv.second));
    return plain;
}

}  // namespace host
}  // namespace iyk
