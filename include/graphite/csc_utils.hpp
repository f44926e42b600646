// forwarding header: the block-sparse Hessian / Schur complement / CSC export live in sparse.hpp (reference path: include/graphite/csc_utils.hpp)
#pragma once
#include "sparse.hpp"
