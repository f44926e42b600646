#!/bin/bash
# ON THE GPU BOX: the GENERIC layer alone (GRAPHITE_GENERIC_ONLY=1: user traits called per factor, stored Jacobians) on a
# 300-camera / 60 000-point / 300 000-factor bundle-adjustment graph, PCG + block-Jacobi: per-iteration wall time from the
# verbose table, rocprofv3 kernel stats, and one PMC pass (FETCH_SIZE / WRITE_SIZE) of the same command.
#   tools/generic_profile.sh TAG [stored|dynamic|auto]
set -u
TAG=${1:-generic}; MODE=${2:-stored}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/generic_$TAG; mkdir -p $OUT
python3 - "$OUT/problem.txt" <<'PY'
import sys
from graphite_amd import synth
synth.write_bal(sys.argv[1], synth.make_problem(300, 60000, 300000, seed=11))
PY
export GRAPHITE_GENERIC_ONLY=1
build/test_generic_bal $OUT/problem.txt pcg 10 $MODE > $OUT/run.out 2>&1
grep -E "^ +[0-9]+ " $OUT/run.out | awk '{n++; t+=$5} END {printf "generic layer, %d LM iterations: %.3f ms per iteration (Time column)\n", n, 1e3*t/n}'
grep -E "FINAL_CHI2|ENGINE_HANDOVERS" $OUT/run.out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- build/test_generic_bal $OUT/problem.txt pcg 10 $MODE > $OUT/prof.out 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- build/test_generic_bal $OUT/problem.txt pcg 3 $MODE > $OUT/pmc_$C.out 2>&1
  find $OUT/pmc_$C -name "*counter_collection.csv" -exec cp {} $OUT/$C.csv \;
  rm -rf $OUT/pmc_$C
done
python3 - "$OUT" <<'PY'
import csv, sys, collections, re, os
out = sys.argv[1]
tot = collections.defaultdict(lambda: [0, 0.0])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, c + ".csv")
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        k = re.sub(r"^void graphite::detail::|<.*", "", r["Kernel_Name"])[:28]
        tot[(k, c)][0] += 1; tot[(k, c)][1] += float(r["Counter_Value"])
    os.remove(p)
with open(os.path.join(out, "pmc_summary.txt"), "w") as f:
    f.write("kernel, counter, launches, mean per launch (KB; FETCH_SIZE counts 64 B per 128-B request on gfx950)\n")
    for (k, c), (n, v) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        f.write("%-30s %-10s %6d %12.1f\n" % (k, c, n, v / n))
print(open(os.path.join(out, "pmc_summary.txt")).read()[:1500])
PY
rm -rf $OUT/stats $OUT/problem.txt
head -14 $OUT/kernel_stats.csv | cut -c1-220
