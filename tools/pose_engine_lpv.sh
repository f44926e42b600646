#!/bin/bash
# pose-graph engine: lanes per vertex by graph size (per-iteration clock of the optimiser's table, second call)
B=build/test_pose_graph
for n in 5000 20000 30000 60000 100000; do
python - $n <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
n = int(sys.argv[1])
p0, fx, e, m, info, tr = synth.make_pose_graph(n)
synth.write_pose_graph('/tmp/g.txt', p0, fx, e, m, info, huber_delta=0.0)
print("POSES", n, "FACTORS", len(e))
PY
for lpv in default 1 2 4 8; do
  if [ $lpv = default ]; then unset GRAPHITE_POSE_LPV; else export GRAPHITE_POSE_LPV=$lpv; fi
  POSE_REPEAT=2 GR_VERBOSE=1 $B /tmp/g.txt pcg 10 manual 10 1.0 2>&1 | awk -v m=$lpv '/REPEAT 1/{p=1} p && /pose-graph engine: set-up/{i=index($0,"vertices x"); d=substr($0, i, 90)} p && NF==6 && $1 ~ /^[0-9]+$/ {if ($1>0) {s+=$5; k++}} END{printf "  lpv %s: %.1f us per LM iteration   [%s]\n", m, 1e6*s/k, d}'
done
done 2>&1 | tee gpurun_out/pose_engine_lpv.txt
