#!/bin/bash
# ON THE GPU BOX: the GENERIC layer's Schur solvers (GRAPHITE_GENERIC_ONLY=1) on a bundle-adjustment graph: per-iteration
# wall time from the verbose table and rocprofv3 kernel stats.
#   tools/generic_schur_profile.sh TAG [workload] [pcg-schur|eigen-schur]
set -u
TAG=${1:-schur}; W=${2:-ladybug-1723}; SOLVER=${3:-pcg-schur}
export TMPDIR=/tmp GRAPHITE_GENERIC_ONLY=1
OUT=$PWD/gpurun_out/generic_$TAG; mkdir -p $OUT
python3 - "$W" "$OUT/problem.txt" <<'PY'
import sys
from graphite_amd import synth
synth.write_bal(sys.argv[2], synth.make_config(sys.argv[1]))
PY
build/test_generic_bal $OUT/problem.txt $SOLVER 6 stored > $OUT/run.out 2>&1
grep -E "^ +[0-9]+ " $OUT/run.out | awk '{n++; t+=$5} END {printf "generic layer, %d LM iterations: %.3f ms per iteration (Time column)\n", n, 1e3*t/n}'
grep -E "FINAL_CHI2|ENGINE_HANDOVERS" $OUT/run.out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- build/test_generic_bal $OUT/problem.txt $SOLVER 6 stored > $OUT/prof.out 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats $OUT/problem.txt
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, re, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= 14: break
    print("%-70s calls %5s avg %10.1f us  %5s %%" % (re.sub(r"graphite::(detail::)?|FactorDescriptor<[^>]*> >, ", "", r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
