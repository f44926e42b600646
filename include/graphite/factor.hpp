// forwarding header: the generic layer lives in core.hpp (reference path: include/graphite/factor.hpp)
#pragma once
#include "core.hpp"
