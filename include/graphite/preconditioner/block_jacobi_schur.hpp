// forwarding header: solvers, preconditioners and the LM driver live in solve.hpp (reference path: include/graphite/preconditioner/block_jacobi_schur.hpp)
#pragma once
#include "../solve.hpp"
