// forwarding header: the generic layer lives in core.hpp (reference path: include/graphite/active.hpp)
#pragma once
#include "core.hpp"
