#!/bin/bash
# phase stamps of the pose-graph engine's last solve for a graph of $1 poses
python - $1 <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
n = int(sys.argv[1])
p0, fx, e, m, info, tr = synth.make_pose_graph(n)
synth.write_pose_graph('/tmp/g.txt', p0, fx, e, m, info, huber_delta=0.0)
PY
POSE_REPEAT=2 GRAPHITE_POSE_DEBUG=1 GR_VERBOSE=1 build/test_pose_graph /tmp/g.txt pcg 6 manual 10 1.0 2>&1 | grep "last solve\|set-up\|operator phase" | tail -3 | cut -c1-700
