// forwarding header: the generic layer lives in core.hpp (reference path: include/graphite/utils.hpp)
#pragma once
#include "core.hpp"
