// graphite/types.hpp (reference path): storage precisions.  The types live in core.hpp; the CUDA names the reference's
// drivers use for them (examples/bal.cu:35, :338-345) are provided here.
#pragma once
#include "core.hpp"
using __nv_bfloat16 = graphite::bfloat16;
