#pragma once
#include "../cereal.hpp"
namespace cereal {
class PortableBinaryOutputArchive : public ArchiveShim<PortableBinaryOutputArchive, false> {
public:
    explicit PortableBinaryOutputArchive(std::ostream&) {}
};
class PortableBinaryInputArchive : public ArchiveShim<PortableBinaryInputArchive, true> {
public:
    explicit PortableBinaryInputArchive(std::istream&) {}
};
}  // namespace cereal
