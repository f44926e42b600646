#!/bin/bash
# EigenLDLTSolver on pose graphs: sparse tile Cholesky (gr_spchol) against the dense MFMA Cholesky, per LM iteration (table's Time column)
for n in 2000 5000 10000 30000; do
python - $n <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
n = int(sys.argv[1])
p0, fx, e, m, info, tr = synth.make_pose_graph(n)
synth.write_pose_graph('/tmp/g.txt', p0, fx, e, m, info, huber_delta=0.0)
print("POSES", n, "FACTORS", len(e))
PY
for mode in sparse dense; do
  if [ $mode = dense ]; then export GRAPHITE_LDLT_SPARSE_MIN=100000000; if [ $n -gt 10000 ]; then continue; fi; else unset GRAPHITE_LDLT_SPARSE_MIN; fi
  build/test_pose_graph /tmp/g.txt eigen 5 manual 10 1.0 2>&1 | awk -v m=$mode 'NF==6 && $1 ~ /^[0-9]+$/ {if ($1>0) {s+=$5; k++}} /SPARSE_FACT/{sp=$2} /FINAL_CHI2/{c=$2} END{printf "  %s (SPARSE_FACTORISATION %s): %.2f ms per LM iteration, final chi2 %s\n", m, sp, 1e3*s/k, c}'
done
done
