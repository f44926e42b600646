#!/bin/bash
# rocprofv3 kernel stats of the Ladybug-49 fp32 Schur + PCG line (BASELINE configs[1])
OUT=$PWD/gpurun_out/lb49; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o s -- python3 /root/repo/bench.py --workload ladybug-49 --solver pcg-schur --dtype f32 --no-also --no-cpu-baseline --repeats 3 > $OUT/bench.json 2> $OUT/err.txt
cd /root/repo
cp $OUT/p/s_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/p
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/lb49/kernel_stats.csv')))
for r in rows[:12]: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
