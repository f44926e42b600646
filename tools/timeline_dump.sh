#!/bin/bash
# kernel-by-kernel trace of a few LM iterations of the default bench line (start offset, duration, gap before)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --no-cpu-baseline --no-also "$@" > gpurun_out/timeline.log 2>&1
find gpurun_out/tl -name "*kernel_trace.csv" -exec cp {} gpurun_out/timeline.csv \;
rm -rf gpurun_out/tl
python3 - <<'PY'
import csv, re
rows = list(csv.DictReader(open("gpurun_out/timeline.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void gr::|^gr::|<.*|\(.*", "", r["Kernel_Name"])[:24]) for r in rows)
calls, cur = [], [ev[0]]
for p, q in zip(ev[:-1], ev[1:]):
    if q[0] - p[1] > 300e3: calls.append(cur); cur = []
    cur.append(q)
calls.append(cur)
calls = [c for c in calls if sum(1 for e in c if e[2] == "k_linearize") >= 15]
c = calls[-2]
t0 = c[0][0]
prev = None
for e in c[:70]:
    print("%9.2f  %-24s dur %7.2f  gap %6.2f" % ((e[0] - t0) / 1e3, e[2], (e[1] - e[0]) / 1e3, 0 if prev is None else (e[0] - prev) / 1e3))
    prev = e[1]
PY
rm -f gpurun_out/timeline.csv
