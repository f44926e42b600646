#!/usr/bin/env python3
"""print the per-kernel table of a bench.py --dump-kernels file"""
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ms"])[: int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    n = max(v.get("active_launches", v["launches"]), 1)
    print("%-22s launches %5d avg %8.1f us  total %8.2f ms" % (k, v["launches"], v["total_ms"] * 1e3 / n, v["total_ms"]))
print("ms_per_step", d["line"]["ms_per_step"], "value", d["line"]["value"])
