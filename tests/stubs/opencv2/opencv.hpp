// TEST-ONLY: the umbrella header the reference's base_extractor.h includes; see core.hpp.
#pragma once
#include <algorithm>
#include <cmath>
#include "core.hpp"
