#!/bin/bash
# ON THE GPU BOX: the reference's examples/bal.cu, compiled UNMODIFIED (build/ref_examples/bal), on a Ladybug-1723-shape BAL
# file; prints which path ran, the per-iteration time of its verbose table and its MSE next to the engine's own bench
# line; rocprofv3 kernel stats of the same command land in gpurun_out/dropin/.
#   tools/dropin_bal.sh [workload] [iterations]
set -u
W=${1:-ladybug-1723}; IT=${2:-20}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/dropin; mkdir -p $OUT
python3 - "$W" "$OUT/problem.txt" <<'PY'
import sys
from graphite_amd import synth
synth.write_bal(sys.argv[2], synth.make_config(sys.argv[1]))
PY
for P in FP64-FP64 FP64-FP32; do
  GR_VERBOSE=1 build/ref_examples/bal $OUT/problem.txt --solver pcg --precision $P --iterations $IT --verbose > $OUT/bal_$P.out 2> $OUT/bal_$P.err
  echo "== bal.cu --solver pcg --precision $P: $(grep -c 'handed to the gr_bal engine' $OUT/bal_$P.err) hand-over(s)"
  grep -E "hand-over probe|handed to" $OUT/bal_$P.err
  grep -E "^ +$((IT-1)) |Optimization took|^MSE" $OUT/bal_$P.out
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- build/ref_examples/bal $OUT/problem.txt --solver pcg --iterations $IT > $OUT/bal_prof.out 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats $OUT/problem.txt
head -12 $OUT/kernel_stats.csv
