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
#include <typeindex>
#include <typeinfo>
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


class PortableBinaryOutputArchive;
class PortableBinaryInputArchive;
namespace exec_detail {
// one binding per CEREAL_REGISTER_TYPE: how to save / load an object of dynamic type D through a void* to the MOST-DERIVED object
struct Binding {
    std::string name;
    void (*save)(PortableBinaryOutputArchive&, const void* mostDerived);
    std::shared_ptr<void> (*load)(PortableBinaryInputArchive&);   // points at the D object
    void (*raise)(void* mostDerived);                              // throw static_cast<D*>(p): the catch site up-casts
};
inline std::map<std::type_index, Binding>& bindingsByType()
{
    static std::map<std::type_index, Binding> m;
    return m;
}
inline std::map<std::string, Binding>& bindingsByName()
{
    static std::map<std::string, Binding> m;
    return m;
}
template <class D>
struct Registrar {
    explicit Registrar(const char* name);
};
}  // namespace exec_detail

class PortableBinaryOutputArchive {
    std::ostream& os_;
    std::map<const void*, uint32_t> seen_;
    std::map<std::string, uint32_t> polyNames_;

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
            if constexpr (std::is_polymorphic_v<E>) {
                if (!v) {
                    const uint32_t id = 0;
                    raw(&id, 4);
                    return;
                }
                const std::type_info& dyn = typeid(*v);
                if (dyn == typeid(E)) {   // the static type itself: bit 30, then the plain pointer wrapper
                    if constexpr (!std::is_abstract_v<E>) {
                        const uint32_t staticType = 0x40000000u;
                        raw(&staticType, 4);
                        pointerWrapper(static_cast<const std::remove_const_t<E>*>(v.get()));
                        return;
                    }
                }
                auto b = bindingsByType().find(std::type_index(dyn));
                if (b == bindingsByType().end())
                    throw Exception(std::string("cereal stand-in: unregistered polymorphic type ") + dyn.name());
                uint32_t nameId;
                auto n = polyNames_.find(b->second.name);
                if (n == polyNames_.end()) {
                    nameId = static_cast<uint32_t>(polyNames_.size()) + 1;
                    polyNames_.emplace(b->second.name, nameId);
                    const uint32_t tagged = nameId | 0x80000000u;
                    raw(&tagged, 4);
                    one(b->second.name);
                }
                else {
                    nameId = n->second;
                    raw(&nameId, 4);
                }
                b->second.save(*this, dynamic_cast<const void*>(v.get()));
            }
            else
                pointerWrapper(v.get());
        }
        else if constexpr (is_weak<T>::value) {
            const std::shared_ptr<typename T::element_type> locked = v.lock();
            one(locked);
        }
        else if constexpr (is_unique<T>::value || is_bitset<T>::value) notModelled("unique_ptr / bitset");
        else if constexpr (has_serialize<T, PortableBinaryOutputArchive>::value) const_cast<T&>(v).serialize(*this);
        else if constexpr (has_save<T, PortableBinaryOutputArchive>::value) v.save(*this);
        else notModelled("a type without serialize / save");
    }
    template <class B>
    void one(const base_class<B>& b) { one(*b.ptr); }

public:
    // cereal's ptr_wrapper: object id (msb set the first time, then the object follows); p is the most-derived object of type D
    template <class D>
    void pointerWrapper(const D* p)
    {
        if (!p) {
            const uint32_t id = 0;
            raw(&id, 4);
            return;
        }
        auto it = seen_.find(static_cast<const void*>(p));
        if (it != seen_.end()) {
            raw(&it->second, 4);
            return;
        }
        const uint32_t id = static_cast<uint32_t>(seen_.size()) + 1;
        seen_.emplace(static_cast<const void*>(p), id);
        const uint32_t tagged = id | 0x80000000u;
        raw(&tagged, 4);
        one(*p);
    }

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
    std::map<uint32_t, std::shared_ptr<void>> seen_;   // keeps every loaded object alive until the archive goes (weak_ptr targets)
    std::map<uint32_t, std::string> polyNames_;

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
            using M = std::remove_const_t<E>;
            if constexpr (std::is_polymorphic_v<E>) {
                uint32_t nameId;
                raw(&nameId, 4);
                if (nameId == 0) {
                    v.reset();
                    return;
                }
                if (nameId & 0x40000000u) {   // the static type itself
                    if constexpr (!std::is_abstract_v<E> && std::is_default_constructible_v<M>)
                        v = std::static_pointer_cast<M>(pointerWrapper<M>());
                    else
                        throw Exception("cereal stand-in: static-type pointer to a type that cannot be constructed");
                    return;
                }
                std::string name;
                if (nameId & 0x80000000u) {
                    one(name);
                    polyNames_[nameId & 0x7fffffffu] = name;
                }
                else {
                    auto n = polyNames_.find(nameId);
                    if (n == polyNames_.end())
                        throw Exception("cereal stand-in: unknown polymorphic name id");
                    name = n->second;
                }
                auto b = bindingsByName().find(name);
                if (b == bindingsByName().end())
                    throw Exception("cereal stand-in: no binding for polymorphic type " + name);
                std::shared_ptr<void> obj = b->second.load(*this);
                if (!obj) {
                    v.reset();
                    return;
                }
                try {
                    b->second.raise(obj.get());
                }
                catch (M* up) {   // derived-to-base conversion done by the handler match (multiple inheritance included)
                    v = std::shared_ptr<E>(obj, up);
                    return;
                }
                throw Exception("cereal stand-in: " + name + " is not derived from the pointer's static type");
            }
            else {
                std::shared_ptr<void> obj = pointerWrapper<M>();
                v = std::static_pointer_cast<E>(std::static_pointer_cast<M>(obj));
            }
        }
        else if constexpr (is_weak<T>::value) {
            std::shared_ptr<typename T::element_type> locked;
            one(locked);
            v = locked;
        }
        else if constexpr (is_unique<T>::value || is_bitset<T>::value) notModelled("unique_ptr / bitset");
        else if constexpr (has_serialize<T, PortableBinaryInputArchive>::value) v.serialize(*this);
        else if constexpr (has_load<T, PortableBinaryInputArchive>::value) v.load(*this);
        else notModelled("a type without serialize / load");
    }
    template <class B>
    void one(base_class<B>& b) { one(*b.ptr); }
    template <class B>
    void one(base_class<B>&& b) { one(*b.ptr); }

public:
    // cereal's ptr_wrapper, reading: a new object is registered BEFORE its contents are read (cycles through weak_ptr resolve)
    template <class D>
    std::shared_ptr<void> pointerWrapper()
    {
        uint32_t id;
        raw(&id, 4);
        if (id == 0) return nullptr;
        if (id & 0x80000000u) {
            std::shared_ptr<D> p(new D());
            seen_[id & 0x7fffffffu] = p;
            one(*p);
            return p;
        }
        auto it = seen_.find(id);
        if (it == seen_.end())
            throw Exception("cereal stand-in: unknown pointer id");
        return it->second;
    }

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
namespace cereal::exec_detail {
template <class D>
Registrar<D>::Registrar(const char* name)
{
    static_assert(std::is_polymorphic_v<D>, "CEREAL_REGISTER_TYPE is for polymorphic types");
    Binding b;
    b.name = name;
    b.save = [](PortableBinaryOutputArchive& ar, const void* p) { ar.pointerWrapper(static_cast<const D*>(p)); };
    b.load = [](PortableBinaryInputArchive& ar) -> std::shared_ptr<void> {
        if constexpr (std::is_default_constructible_v<D> && !std::is_abstract_v<D>) return ar.template pointerWrapper<D>();
        else notModelled("loading a registered type without a default constructor");
    };
    b.raise = [](void* p) { throw static_cast<D*>(p); };
    bindingsByType().emplace(std::type_index(typeid(D)), b);
    bindingsByName().emplace(b.name, b);
}
}  // namespace cereal::exec_detail
// cereal binds the type to its SPELLED name (#T) in every translation unit that sees the line; so does this (the maps ignore repeats)
#define CEREAL_REGISTER_TYPE(...)                                                                                  \
    namespace cereal_exec_registered {                                                                             \
    inline const ::cereal::exec_detail::Registrar<__VA_ARGS__> CEREAL_EXEC_CAT(reg_, __COUNTER__){#__VA_ARGS__};   \
    }
#define CEREAL_REGISTER_POLYMORPHIC_RELATION(...)
#define CEREAL_CLASS_VERSION(...)
