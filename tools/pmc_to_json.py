#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into profiles/pmc_traffic.json.

usage: pmc_to_json.py KEY FETCH_CSV WRITE_CSV [OUT_JSON]
KEY is "<workload> <dtype> <solver>" as bench.py builds it.  Per kernel the MEDIAN over launches is kept
(early-exit launches of the persistent PCG kernels touch almost nothing and would drag a mean down).
hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024: gfx950 FETCH_SIZE counts 64 B per 128-B request
(MI355X_MICROARCH.md, HBM/rocprofv3 section), WRITE_SIZE is in KB of 64-B writes and needs no correction.
"""
import csv, json, re, statistics, sys
from collections import defaultdict


def short(name):
    m = re.search(r"(?:gr::)?(k_\w+|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:40]


def load(path, counter):
    per = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return per


def main():
    key, fcsv, wcsv = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_traffic.json"
    f, w = load(fcsv, "FETCH_SIZE"), load(wcsv, "WRITE_SIZE")
    try:
        doc = json.load(open(out))
    except Exception:
        doc = {}
    doc["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, median over launches; "
                   "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE counts 64 B per 128-B request; "
                   "uncalibrated for the 24..192-byte gathers of these kernels, so the x2 is an upper bound)")
    ent = {}
    for k in sorted(set(f) | set(w)):
        fk = statistics.median(f[k]) if f.get(k) else 0.0
        wk = statistics.median(w[k]) if w.get(k) else 0.0
        ent[k] = {"hbm_bytes": 2 * fk * 1024 + wk * 1024, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                  "launches": len(f.get(k, []))}
    doc[key] = ent
    json.dump(doc, open(out, "w"), indent=1)
    print("wrote", out, key, len(ent), "kernels")


if __name__ == "__main__":
    main()
