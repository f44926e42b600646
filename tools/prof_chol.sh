#!/bin/bash
# usage (on the GPU box): tools/prof_chol.sh TAG sizes...   -> gpurun_out/chol_stats_TAG.csv
TAG=$1; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chol_$TAG -o c -- python3 tools/chol_bench.py "$@" > gpurun_out/chol_prof_$TAG.log 2>&1
find gpurun_out/prof_chol_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/chol_stats_$TAG.csv \;
rm -rf gpurun_out/prof_chol_$TAG
python3 - gpurun_out/chol_stats_$TAG.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_chol' in r['Name']:
        print("%-46s %5s calls total %9.3f ms avg %9.1f us min %8.1f max %9.1f" % (r['Name'][:46], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
