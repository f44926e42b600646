// forwarding header: the generic layer lives in core.hpp (reference path: include/graphite/types.hpp)
#pragma once
#include "core.hpp"
