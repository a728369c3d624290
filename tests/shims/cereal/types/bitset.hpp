// compile-only shim (tests/shims/README.md): like the real header, brings in the standard type it serialises
#pragma once
#include <bitset>
#include "../cereal.hpp"
