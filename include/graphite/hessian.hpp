// forwarding header: the block-sparse Hessian / Schur complement / CSC export live in sparse.hpp (reference path: include/graphite/hessian.hpp)
#pragma once
#include "sparse.hpp"
