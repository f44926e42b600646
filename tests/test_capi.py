"""The C-ABI library loads on a CPU-only machine, exports every symbol include/graphite_mi355x.h
declares, and FAILS LOUDLY (no CPU fallback) when a compute entry point is used without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from graphite_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    # the drop-in boundary: graphite_mi355x.h + its extension for user-traits problems, graphite_mi355x_model.h
    syms = set()
    for h in ("graphite_mi355x.h", "graphite_mi355x_model.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"^\s*(?:gr_status|const char \*|int|void)\s*(gr_[a-z0-9_]+)\s*\(", text, flags=re.M))
    return sorted(syms)


def test_library_builds_and_exports_every_declared_symbol():
    _lib.build()
    lib = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == syms
    assert b"gfx950" in lib.gr_version()
    # test / diagnostic entry points live in their own header, outside the drop-in boundary
    test_text = open(os.path.join(ROOT, "include", "graphite_mi355x_test.h")).read()
    for s in _lib.TEST_EXPORTS:
        assert s in test_text and s not in syms and hasattr(lib, s)


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "graphite_mi355x.h"\nint main(void){gr_lm_options o; (void)o; return sizeof(gr_kernel_stat) > 0 ? 0 : 1;}\n')
    assert os.system(f"gcc -std=c99 -Wall -Werror -I{ROOT}/include -c {src} -o {tmp_path}/t.o") == 0


def test_struct_layouts_match_ctypes(tmp_path):
    src = tmp_path / "s.c"
    src.write_text('#include <stdio.h>\n#include "graphite_mi355x_model.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(gr_lm_options), sizeof(gr_lm_stats), sizeof(gr_kernel_stat), sizeof(gr_bal_tuning), sizeof(gr_comm_info));return 0;}\n')
    assert os.system(f"gcc -I{ROOT}/include {src} -o {tmp_path}/s") == 0
    sizes = [int(x) for x in os.popen(f"{tmp_path}/s").read().split()]
    assert sizes == [C.sizeof(_lib.LMOptions), C.sizeof(_lib.LMStats), C.sizeof(_lib.KernelStat), C.sizeof(_lib.Tuning), C.sizeof(_lib.CommInfo)]
    assert sizes[3] == 100  # gr_version 0.2: new tuning fields take `reserved` slots


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.lib()
    assert lib.gr_device_count() == 0
    prob = synth.make_config("mini-6")
    import graphite_amd as ga
    with pytest.raises(_lib.GraphiteError) as e:
        ga.BalProblem(prob.cameras, prob.points, prob.obs, prob.cam_idx, prob.pt_idx)
    assert e.value.status == 3  # GR_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under graphite_amd/ may import, link or execute it."""
    pkg = os.path.join(ROOT, "graphite_amd")
    pat = re.compile(r"^\s*(import\s+oracle|from\s+oracle|#include\s+\"[^\"]*oracle)", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f
                assert "libgraphite_oracle" not in text, f
