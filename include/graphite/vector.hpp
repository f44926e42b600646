// forwarding header: the generic layer lives in core.hpp (reference path: include/graphite/vector.hpp)
#pragma once
#include "core.hpp"
