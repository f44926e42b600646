// forwarding header: solvers, preconditioners and the LM driver live in solve.hpp (reference path: include/graphite/solver/eigen_schur.hpp)
#pragma once
#include "../solve.hpp"
