#!/bin/bash
# pose-graph engine: feature checks by hand (output under gpurun_out/)
mkdir -p gpurun_out
python - <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
p0, fx, e, m, info, tr = synth.make_pose_graph(2000)
synth.write_pose_graph('/tmp/g2k.txt', p0, fx, e, m, info, huber_delta=0.0)
PY
{
B=build/test_pose_graph
echo "== f32 engine"; $B /tmp/g2k.txt pcg 8 manual-f32 10 1.0 | grep -E "^ +[0-9]+ |FINAL|HANDOVERS"
echo "== f32 generic"; GRAPHITE_GENERIC_ONLY=1 $B /tmp/g2k.txt pcg 8 manual-f32 10 1.0 | grep -E "^ +[0-9]+ |FINAL|HANDOVERS"
echo "== lm2 engine"; POSE_LM2=1 $B /tmp/g2k.txt pcg 30 manual 10 1.0 | grep -E "^ +[0-9]+ |FINAL|HANDOVERS"
echo "== lm2 generic"; POSE_LM2=1 GRAPHITE_GENERIC_ONLY=1 $B /tmp/g2k.txt pcg 30 manual 10 1.0 | grep -E "^ +[0-9]+ |FINAL|HANDOVERS"
echo "== stop engine"; POSE_STOP=1 $B /tmp/g2k.txt pcg 8 manual 10 1.0 | grep -E "^ +[0-9]+ |FINAL|HANDOVERS|Stopping"
echo "== stop generic"; POSE_STOP=1 GRAPHITE_GENERIC_ONLY=1 $B /tmp/g2k.txt pcg 8 manual 10 1.0 | grep -E "^ +[0-9]+ |FINAL|HANDOVERS|Stopping"
echo "== vec6 engine"; $B vec6 3000 8 x /tmp/v6a.txt | grep -E "^ +[0-9]+ |FINAL|HANDOVERS|POSES"
echo "== vec6 generic"; GRAPHITE_POSE_ENGINE=0 $B vec6 3000 8 x /tmp/v6b.txt | grep -E "^ +[0-9]+ |FINAL|HANDOVERS"
python - <<'PY'
import numpy as np
a, b = np.loadtxt('/tmp/v6a.txt'), np.loadtxt('/tmp/v6b.txt')
print("vec6 max |engine - generic|", np.abs(a - b).max())
PY
echo "== timeout + fallback"; POSE_REPEAT=2 GR_VERBOSE=1 GRAPHITE_POSE_VAR=256 $B /tmp/g2k.txt pcg 8 manual 10 1.0 2>&1 | grep -E "^ +(0|7) |FINAL|HANDOVERS|pose-graph engine|REPEAT"
} > gpurun_out/pose_engine_try.txt 2>&1
cut -c1-400 gpurun_out/pose_engine_try.txt
