// tests/upstream_exec (README.md there): a small WORKING subset of cereal — PortableBinary{Input,Output}Archive over arithmetic and
// enum values, std::string, vector, array, pair, tuple, (unordered_)map, optional and non-polymorphic shared_ptr, classes through
// their serialize / save / load members, cereal::base_class — written from cereal's published encoding rules (the same rules
// iyokan_amd/packet.py and iyokan_amd/host/packet.hpp restate): 1 byte "little endian" flag, size tags of 8 bytes, containers as
// size tag + elements, std::array as its elements, optional as a `nullopt` byte + the value, shared_ptr as a 32-bit id (msb set: the
// object follows).  Enough to run upstream's readFromArchive / writeToArchive on PlainPacket, TFHEPacket and the stand-in EvalKey.
// POLYMORPHIC pointers (upstream's snapshots: CEREAL_REGISTER_TYPE) and weak_ptr compile and abort when reached: not modelled.
// Test infrastructure only; cereal itself is not in this container.
#pragma once
#include <array>
#include <bitset>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <istream>
#include <map>
#include <memory>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace cereal {
struct Exception : std::runtime_error {
    using std::runtime_error::runtime_error;
};

template <class Base>
struct base_class {
    Base* ptr;
    template <class Derived>
    base_class(Derived* d) : ptr(static_cast<Base*>(const_cast<std::remove_const_t<Derived>*>(d)))
    {
    }
};
template <class Base>
using virtual_base_class = base_class<Base>;

namespace exec_detail {
[[noreturn]] inline void notModelled(const char* what)
{
    std::fprintf(stderr, "tests/upstream_exec cereal stand-in: %s is not modelled (the executed tests must not reach it)\n", what);
    std::abort();
}
template <class T, class A, class = void> struct has_serialize : std::false_type {};
template <class T, class A> struct has_serialize<T, A, std::void_t<decltype(std::declval<T&>().serialize(std::declval<A&>()))>> : std::true_type {};
template <class T, class A, class = void> struct has_save : std::false_type {};
template <class T, class A> struct has_save<T, A, std::void_t<decltype(std::declval<const T&>().save(std::declval<A&>()))>> : std::true_type {};
template <class T, class A, class = void> struct has_load : std::false_type {};
template <class T, class A> struct has_load<T, A, std::void_t<decltype(std::declval<T&>().load(std::declval<A&>()))>> : std::true_type {};
template <class T> struct is_std_array : std::false_type {};
template <class T, size_t N> struct is_std_array<std::array<T, N>> : std::true_type {};
template <class T> struct is_vector : std::false_type {};
template <class T, class A> struct is_vector<std::vector<T, A>> : std::true_type {};
template <class T> struct is_map : std::false_type {};
template <class K, class V, class H, class E, class A> struct is_map<std::unordered_map<K, V, H, E, A>> : std::true_type {};
template <class K, class V, class C, class A> struct is_map<std::map<K, V, C, A>> : std::true_type {};
template <class T> struct is_optional : std::false_type {};
template <class T> struct is_optional<std::optional<T>> : std::true_type {};
template <class T> struct is_shared : std::false_type {};
template <class T> struct is_shared<std::shared_ptr<T>> : std::true_type {};
template <class T> struct is_weak : std::false_type {};
template <class T> struct is_weak<std::weak_ptr<T>> : std::true_type {};
template <class T> struct is_unique : std::false_type {};
template <class T, class D> struct is_unique<std::unique_ptr<T, D>> : std::true_type {};
template <class T> struct is_pair : std::false_type {};
template <class A, class B> struct is_pair<std::pair<A, B>> : std::true_type {};
template <class T> struct is_tuple : std::false_type {};
template <class... A> struct is_tuple<std::tuple<A...>> : std::true_type {};
template <class T> struct is_bitset : std::false_type {};
template <size_t N> struct is_bitset<std::bitset<N>> : std::true_type {};
}  // namespace exec_detail

class PortableBinaryOutputArchive {
    std::ostream& os_;
    std::map<const void*, uint32_t> seen_;

    void raw(const void* p, size_t n)
    {
        os_.write(static_cast<const char*>(p), static_cast<std::streamsize>(n));
        if (!os_)
            throw Exception("cereal stand-in: write failed");
    }
    void size(uint64_t n) { raw(&n, 8); }

    template <class T>
    void one(const T& v)
    {
        using namespace exec_detail;
        if constexpr (std::is_arithmetic_v<T>) raw(&v, sizeof v);
        else if constexpr (std::is_enum_v<T>) {
            const auto u = static_cast<std::underlying_type_t<T>>(v);
            raw(&u, sizeof u);
        }
        else if constexpr (std::is_same_v<T, std::string>) {
            size(v.size());
            raw(v.data(), v.size());
        }
        else if constexpr (is_vector<T>::value) {
            size(v.size());
            for (const auto& e : v) {
                const typename T::value_type& x = e;   // (vector<bool> proxies become values)
                one(x);
            }
        }
        else if constexpr (is_std_array<T>::value) {
            for (const auto& e : v) one(e);
        }
        else if constexpr (is_map<T>::value) {
            size(v.size());
            for (const auto& kv : v) {
                one(kv.first);
                one(kv.second);
            }
        }
        else if constexpr (is_pair<T>::value) {
            one(v.first);
            one(v.second);
        }
        else if constexpr (is_tuple<T>::value) std::apply([this](const auto&... e) { (one(e), ...); }, v);
        else if constexpr (is_optional<T>::value) {
            const bool nullopt = !v.has_value();
            one(nullopt);
            if (!nullopt) one(*v);
        }
        else if constexpr (is_shared<T>::value) {
            using E = typename T::element_type;
            if constexpr (std::is_polymorphic_v<E>) notModelled("a polymorphic shared_ptr (snapshot)");
            else {
                if (!v) {
                    const uint32_t id = 0;
                    raw(&id, 4);
                    return;
                }
                auto it = seen_.find(v.get());
                if (it != seen_.end()) {
                    raw(&it->second, 4);
                    return;
                }
                const uint32_t id = static_cast<uint32_t>(seen_.size()) + 1;
                seen_.emplace(v.get(), id);
                const uint32_t tagged = id | 0x80000000u;
                raw(&tagged, 4);
                one(*v);
            }
        }
        else if constexpr (is_weak<T>::value || is_unique<T>::value || is_bitset<T>::value) notModelled("weak_ptr / unique_ptr / bitset");
        else if constexpr (has_serialize<T, PortableBinaryOutputArchive>::value) const_cast<T&>(v).serialize(*this);
        else if constexpr (has_save<T, PortableBinaryOutputArchive>::value) v.save(*this);
        else notModelled("a type without serialize / save");
    }
    template <class B>
    void one(const base_class<B>& b) { one(*b.ptr); }

public:
    explicit PortableBinaryOutputArchive(std::ostream& os) : os_(os)
    {
        const uint8_t littleEndian = 1;
        raw(&littleEndian, 1);
    }
    template <class... T>
    PortableBinaryOutputArchive& operator()(T&&... t)
    {
        (one(t), ...);
        return *this;
    }
};

class PortableBinaryInputArchive {
    std::istream& is_;
    std::map<uint32_t, std::shared_ptr<void>> seen_;

    void raw(void* p, size_t n)
    {
        is_.read(static_cast<char*>(p), static_cast<std::streamsize>(n));
        if (static_cast<size_t>(is_.gcount()) != n)
            throw Exception("cereal stand-in: archive ends early");
    }
    uint64_t size()
    {
        uint64_t n;
        raw(&n, 8);
        if (n > (uint64_t(1) << 32))
            throw Exception("cereal stand-in: implausible size tag");
        return n;
    }

    template <class T>
    void one(T& v)
    {
        using namespace exec_detail;
        if constexpr (std::is_arithmetic_v<T>) raw(&v, sizeof v);
        else if constexpr (std::is_enum_v<T>) {
            std::underlying_type_t<T> u;
            raw(&u, sizeof u);
            v = static_cast<T>(u);
        }
        else if constexpr (std::is_same_v<T, std::string>) {
            v.resize(size());
            raw(v.data(), v.size());
        }
        else if constexpr (is_vector<T>::value) {
            v.resize(size());
            for (size_t i = 0; i < v.size(); i++) {
                typename T::value_type x{};
                one(x);
                v[i] = std::move(x);
            }
        }
        else if constexpr (is_std_array<T>::value) {
            for (auto& e : v) one(e);
        }
        else if constexpr (is_map<T>::value) {
            const uint64_t n = size();
            v.clear();
            for (uint64_t i = 0; i < n; i++) {
                typename T::key_type k{};
                typename T::mapped_type m{};
                one(k);
                one(m);
                v.emplace(std::move(k), std::move(m));
            }
        }
        else if constexpr (is_pair<T>::value) {
            one(v.first);
            one(v.second);
        }
        else if constexpr (is_tuple<T>::value) std::apply([this](auto&... e) { (one(e), ...); }, v);
        else if constexpr (is_optional<T>::value) {
            bool nullopt = true;
            one(nullopt);
            if (nullopt) v.reset();
            else {
                typename T::value_type x{};
                one(x);
                v = std::move(x);
            }
        }
        else if constexpr (is_shared<T>::value) {
            using E = typename T::element_type;
            if constexpr (std::is_polymorphic_v<E>) notModelled("a polymorphic shared_ptr (snapshot)");
            else {
                uint32_t id;
                raw(&id, 4);
                if (id == 0) v.reset();
                else if (id & 0x80000000u) {
                    auto p = std::make_shared<std::remove_const_t<E>>();
                    seen_[id & 0x7fffffffu] = p;
                    one(*p);
                    v = p;
                }
                else {
                    auto it = seen_.find(id);
                    if (it == seen_.end())
                        throw Exception("cereal stand-in: unknown pointer id");
                    v = std::static_pointer_cast<E>(it->second);
                }
            }
        }
        else if constexpr (is_weak<T>::value || is_unique<T>::value || is_bitset<T>::value) notModelled("weak_ptr / unique_ptr / bitset");
        else if constexpr (has_serialize<T, PortableBinaryInputArchive>::value) v.serialize(*this);
        else if constexpr (has_load<T, PortableBinaryInputArchive>::value) v.load(*this);
        else notModelled("a type without serialize / load");
    }
    template <class B>
    void one(base_class<B>& b) { one(*b.ptr); }
    template <class B>
    void one(base_class<B>&& b) { one(*b.ptr); }

public:
    explicit PortableBinaryInputArchive(std::istream& is) : is_(is)
    {
        uint8_t flag = 0;
        raw(&flag, 1);
        if (flag != 1)
            throw Exception("cereal stand-in: not a little-endian portable binary archive");
    }
    template <class... T>
    PortableBinaryInputArchive& operator()(T&&... t)
    {
        (one(t), ...);
        return *this;
    }
};
}  // namespace cereal

#define CEREAL_EXEC_CAT2(a, b) a##b
#define CEREAL_EXEC_CAT(a, b) CEREAL_EXEC_CAT2(a, b)
// as in tests/shims: registering a polymorphic type instantiates its serialisation for both archives (type check); nothing runs
#define CEREAL_REGISTER_TYPE(...)                                                                                \
    namespace cereal_exec_registered {                                                                           \
    inline void CEREAL_EXEC_CAT(touch_, __COUNTER__)(__VA_ARGS__ & t, ::cereal::PortableBinaryOutputArchive & o,  \
                                                     ::cereal::PortableBinaryInputArchive & i)                    \
    {                                                                                                            \
        o(t);                                                                                                    \
        i(t);                                                                                                    \
    }                                                                                                            \
    }
#define CEREAL_REGISTER_POLYMORPHIC_RELATION(...)
#define CEREAL_CLASS_VERSION(...)
