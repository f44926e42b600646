// forwarding header: solvers, preconditioners and the LM driver live in solve.hpp (reference path: include/graphite/optimizer/levenberg_marquardt.hpp)
#pragma once
#include "../solve.hpp"
