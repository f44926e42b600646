#!/bin/bash
# pose-graph engine: first-use cost (POSE_REPEAT) and cooperative vs plain launch
python - <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
p0, fx, e, m, info, tr = synth.make_pose_graph(10000)
synth.write_pose_graph('/tmp/g10k.txt', p0, fx, e, m, info, huber_delta=0.0)
PY
mkdir -p gpurun_out
for coop in 1 0; do
  echo "COOP $coop"
  POSE_REPEAT=3 GRAPHITE_POSE_COOP=$coop GR_VERBOSE=1 build/test_pose_graph /tmp/g10k.txt pcg 20 manual 10 1.0 2>&1 | grep -E "^ +(0|1|2|19) |REPEAT|LM_SECONDS|ENGINE_SETUP|pose-graph engine:"
done 2>&1 | tee gpurun_out/pose_engine_var.txt
echo GENERIC
POSE_REPEAT=3 GRAPHITE_GENERIC_ONLY=1 build/test_pose_graph /tmp/g10k.txt pcg 20 manual 10 1.0 2>&1 | grep -E "^ +(0|1|2|19) |REPEAT|LM_SECONDS" | tee -a gpurun_out/pose_engine_var.txt
