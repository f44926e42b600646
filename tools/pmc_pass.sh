#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass of bench.py with the given counters, per-kernel MEDIAN printed.
#   tools/pmc_pass.sh TAG "COUNTER COUNTER ..." [bench.py args...]
set -u
TAG=$1; CNT=$2; shift; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $CNT --output-format csv -d $OUT/p -o p -- python3 bench.py --no-cpu-baseline --no-also --repeats 1 --steps 10 --warmup 2 "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, sys, os, collections, glob, statistics
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k, d in sorted(agg.items()):
        if not any(x in k for x in ("k_linearize", "k_pcg", "k_sp_", "k_chol")): continue
        line = k.ljust(50) + " ".join(f"{c}={statistics.median(v):.4g}" for c, v in sorted(d.items()))
        print(line); fh.write(line + "\n")
PY
rm -rf $OUT/p
