#!/bin/bash
# pose-graph engine against the generic kernels by graph size (second call of the process; per-iteration clock of the optimiser's table)
mkdir -p gpurun_out
B=build/test_pose_graph
for n in 1000 10000 100000 400000; do
python - $n <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
n = int(sys.argv[1])
p0, fx, e, m, info, tr = synth.make_pose_graph(n)
synth.write_pose_graph('/tmp/g.txt', p0, fx, e, m, info, huber_delta=0.0)
print("POSES", n, "FACTORS", len(e))
PY
for mode in engine generic; do
  if [ $mode = generic ]; then export GRAPHITE_GENERIC_ONLY=1; else unset GRAPHITE_GENERIC_ONLY; fi
  POSE_REPEAT=2 GR_VERBOSE=1 $B /tmp/g.txt pcg 10 manual 10 1.0 2>&1 | awk -v m=$mode '/REPEAT 1/{p=1} p && /pose-graph engine: set-up/{print "  " m ": " substr($0, index($0,"vertices")-8, 120)} p && NF==6 && $1 ~ /^[0-9]+$/ {n++; if ($1>0) {s+=$5; k++}} /LM_SECONDS/{w=$2} END{printf "  %s: %.1f us per LM iteration, whole call %.2f ms\n", m, 1e6*s/k, 1e3*w}'
done
done 2>&1 | tee gpurun_out/pose_engine_scale.txt
