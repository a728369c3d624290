#pragma once
#include "../cereal.hpp"
