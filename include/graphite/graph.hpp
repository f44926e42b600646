// forwarding header: the generic layer lives in core.hpp (reference path: include/graphite/graph.hpp)
#pragma once
#include "core.hpp"
