#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel stats + the two PMC passes for one
# bench.py configuration; everything lands under gpurun_out/prof_<tag>/ for copying into profiles/.
#   tools/collect_profiles.sh TAG [bench.py args...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--no-cpu-baseline --steps 20 --warmup 3 $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py $ARGS > $OUT/bench_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 bench.py $ARGS > $OUT/bench_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 bench.py $ARGS > $OUT/bench_write.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/fetch -name "*counter_collection.csv" -exec cp {} $OUT/fetch.csv \;
find $OUT/write -name "*counter_collection.csv" -exec cp {} $OUT/write.csv \;
rm -rf $OUT/stats $OUT/fetch $OUT/write
# counter CSVs are large: keep one row per (kernel, value) summary only
python3 - "$OUT" <<'PY'
import csv, sys, statistics, collections, os
out = sys.argv[1]
for f in ("fetch", "write"):
    p = os.path.join(out, f + ".csv")
    if not os.path.exists(p): continue
    rows = list(csv.DictReader(open(p)))
    keep = ["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
    with open(p, "w", newline="") as fh:
        w = csv.DictWriter(fh, keep, quoting=csv.QUOTE_NONNUMERIC); w.writeheader()
        for r in rows: w.writerow({k: r[k] for k in keep})
PY
ls -la $OUT
