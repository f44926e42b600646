#!/bin/bash
# ON THE GPU BOX: the pose-graph engine (include/graphite/engine_pose.hpp) on the 10 000-pose / 48 593-factor SE(2) graph of
# graphite_amd.synth.make_pose_graph, fp64, PCGSolver + block-Jacobi, 10 PCG iterations per LM iteration, 20 LM iterations:
# per-iteration times of the optimiser's table (second call of the process), the solve's phase stamps, rocprofv3 kernel stats of
# the engine and of the generic kernels on the same command.  Output: gpurun_out/pose_prof/ (copy the summaries into profiles/).
OUT=$PWD/gpurun_out/pose_prof; rm -rf $OUT; mkdir -p $OUT
python - <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
p0, fx, e, m, info, tr = synth.make_pose_graph(10000)
synth.write_pose_graph('/tmp/g10k.txt', p0, fx, e, m, info, huber_delta=0.0)
PY
B=$PWD/build/test_pose_graph
ARGS="/tmp/g10k.txt pcg 20 manual 10 1.0"
{
echo "# pose-graph engine, 10 000 poses / 48 593 factors, fp64, PCG + block-Jacobi, 10 PCG iterations per LM iteration"
echo "## engine, second call of the process (POSE_REPEAT=2), GR_VERBOSE + GRAPHITE_POSE_DEBUG"
POSE_REPEAT=2 GRAPHITE_POSE_DEBUG=1 GR_VERBOSE=1 $B $ARGS 2>&1 | grep -v "hand-over:" | awk '/REPEAT 1/{p=1} p'
echo "## generic kernels (GRAPHITE_GENERIC_ONLY=1), second call of the process"
POSE_REPEAT=2 GRAPHITE_GENERIC_ONLY=1 $B $ARGS 2>&1 | awk '/REPEAT 1/{p=1} p'
} > $OUT/tables.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e -o pe -- $B $ARGS > $OUT/engine_prof.log 2>&1
GRAPHITE_GENERIC_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/g -o pg -- $B $ARGS > $OUT/generic_prof.log 2>&1
cd - > /dev/null
cp $OUT/e/pe_kernel_stats.csv $OUT/engine_kernel_stats.csv; cp $OUT/g/pg_kernel_stats.csv $OUT/generic_kernel_stats.csv
rm -rf $OUT/e $OUT/g
grep -E "ENGINE_LOOP|LM_SECONDS|last solve" $OUT/tables.txt | cut -c1-300
head -5 $OUT/engine_kernel_stats.csv | cut -c1-150
