#!/bin/bash
# rocprofv3 kernel stats of EigenLDLTSolver (sparse tile Cholesky) on the 10 k-pose graph
OUT=$PWD/gpurun_out/pose_direct; rm -rf $OUT; mkdir -p $OUT
python - <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
p0, fx, e, m, info, tr = synth.make_pose_graph(10000)
synth.write_pose_graph('/tmp/g10k.txt', p0, fx, e, m, info, huber_delta=0.0)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o s -- /root/repo/build/test_pose_graph /tmp/g10k.txt eigen 10 manual 10 1.0 > $OUT/run.log 2>&1
cd /root/repo
cp $OUT/p/s_kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/p
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/pose_direct/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6, "launches", sum(int(r['Calls']) for r in rows))
for r in rows[:14]: print(r['Name'][:64], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(100*float(r['TotalDurationNs'])/tot,1))
PY
grep -E "^ +(1|2|9) " $OUT/run.log | cut -c1-140
