// forwarding header: the generic layer lives in core.hpp (reference path: include/graphite/stream.hpp)
#pragma once
#include "core.hpp"
