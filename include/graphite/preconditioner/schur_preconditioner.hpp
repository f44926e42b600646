// forwarding header: solvers, preconditioners and the LM driver live in solve.hpp (reference path: include/graphite/preconditioner/schur_preconditioner.hpp)
#pragma once
#include "../solve.hpp"
