// compile-only shim (tests/shims/README.md): cereal as /root/reference/src uses it — ar(...), cereal::base_class<B>(this),
// CEREAL_REGISTER_TYPE(T), PortableBinary{Input,Output}Archive.  An archive's operator() visits every argument that has a
// serialize / save / load member, so that the serialize templates of upstream's and the plugin's classes are INSTANTIATED
// (type-checked); nothing is read or written.
#pragma once
#include <cstdint>
#include <istream>
#include <ostream>
#include <type_traits>
#include <utility>
namespace cereal {
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

namespace shim {
template <class T, class A, class = void>
struct has_member_serialize : std::false_type {
};
template <class T, class A>
struct has_member_serialize<T, A, std::void_t<decltype(std::declval<T&>().serialize(std::declval<A&>()))>> : std::true_type {
};
template <class T, class A, class = void>
struct has_member_save : std::false_type {
};
template <class T, class A>
struct has_member_save<T, A, std::void_t<decltype(std::declval<const T&>().save(std::declval<A&>()))>> : std::true_type {
};
template <class T, class A, class = void>
struct has_member_load : std::false_type {
};
template <class T, class A>
struct has_member_load<T, A, std::void_t<decltype(std::declval<T&>().load(std::declval<A&>()))>> : std::true_type {
};
}  // namespace shim

template <class Derived, bool IsInput>
class ArchiveShim {
    Derived& self()
    {
        return static_cast<Derived&>(*this);
    }
    template <class B>
    void visit(base_class<B> b)
    {
        visit(*b.ptr);
    }
    template <class T>
    void visit(T&& t)
    {
        using U = std::remove_cv_t<std::remove_reference_t<T>>;
        if constexpr (shim::has_member_serialize<U, Derived>::value)
            const_cast<U&>(t).serialize(self());
        else if constexpr (IsInput && shim::has_member_load<U, Derived>::value)
            const_cast<U&>(t).load(self());
        else if constexpr (!IsInput && shim::has_member_save<U, Derived>::value)
            t.save(self());
    }

public:
    template <class... T>
    Derived& operator()(T&&... t)
    {
        (visit(std::forward<T>(t)), ...);
        return self();
    }
};
}  // namespace cereal

#include "archives/portable_binary.hpp"

#define CEREAL_SHIM_CAT2(a, b) a##b
#define CEREAL_SHIM_CAT(a, b) CEREAL_SHIM_CAT2(a, b)
// registering a polymorphic type makes cereal instantiate its serialisation for every registered archive: the shim instantiates
// it for the two PortableBinary archives
#define CEREAL_REGISTER_TYPE(...)                                                                               \
    namespace cereal_shim_registered {                                                                          \
    inline void CEREAL_SHIM_CAT(touch_, __COUNTER__)(__VA_ARGS__ & t, ::cereal::PortableBinaryOutputArchive & o, \
                                                     ::cereal::PortableBinaryInputArchive & i)                   \
    {                                                                                                           \
        o(t);                                                                                                   \
        i(t);                                                                                                   \
    }                                                                                                           \
    }
#define CEREAL_REGISTER_POLYMORPHIC_RELATION(...)
#define CEREAL_CLASS_VERSION(...)
