#pragma once
#include "format.h"
