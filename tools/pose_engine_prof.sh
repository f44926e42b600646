#!/bin/bash
# rocprofv3 kernel stats of the pose-graph engine on the 10 k-pose graph + its phase stamps; output under gpurun_out/
mkdir -p gpurun_out/pose_prof
python - <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
p0, fx, e, m, info, tr = synth.make_pose_graph(10000)
synth.write_pose_graph('/tmp/g10k.txt', p0, fx, e, m, info, huber_delta=0.0)
PY
GRAPHITE_POSE_DEBUG=1 GR_VERBOSE=1 build/test_pose_graph /tmp/g10k.txt pcg 20 manual 10 1.0 2>&1 | grep "pose-graph engine" > gpurun_out/pose_prof/stamps.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/pose_prof -o pe -- /root/repo/build/test_pose_graph /tmp/g10k.txt pcg 20 manual 10 1.0 > /root/repo/gpurun_out/pose_prof/run.log 2>&1
cd /root/repo
cat gpurun_out/pose_prof/stamps.txt
find gpurun_out/pose_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {} | cut -c1-220
