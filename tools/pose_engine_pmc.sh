#!/bin/bash
# HBM-side traffic of the pose-graph engine's kernels on the 10 k-pose graph: FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes
OUT=$PWD/gpurun_out/pose_pmc; rm -rf $OUT; mkdir -p $OUT
python - <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
p0, fx, e, m, info, tr = synth.make_pose_graph(10000)
synth.write_pose_graph('/tmp/g10k.txt', p0, fx, e, m, info, huber_delta=0.0)
PY
B=$PWD/build/test_pose_graph
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o p -- $B /tmp/g10k.txt pcg 10 manual 10 1.0 > $OUT/$C.log 2>&1
  find $OUT/$C -name "*counter_collection.csv" -exec cp {} $OUT/$C.csv \;
  rm -rf $OUT/$C
done
cd - > /dev/null
python3 - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{c}.csv")):
        name = r["Kernel_Name"].split("<")[0].split("::")[-1]
        tot[name][c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE": cnt[name] += 1
print("kernel, launches, FETCH_SIZE per launch (counter units of 64 B... see MI355X_MICROARCH.md), WRITE_SIZE per launch")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]["FETCH_SIZE"])[:6]:
    n = max(cnt[k], 1)
    print(f"{k:28s} {n:4d}  fetch {v['FETCH_SIZE'] / n:12.1f}  write {v['WRITE_SIZE'] / n:12.1f}")
PY
