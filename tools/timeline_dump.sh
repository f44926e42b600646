#!/bin/bash
# kernel-by-kernel trace of the first LM iterations of a bench.py configuration (start offset, duration, gap before)
#   tools/timeline_dump.sh [N kernels] [bench.py args...]
export TMPDIR=/tmp
N=${1:-70}; shift
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --no-cpu-baseline --no-also "$@" > gpurun_out/timeline.log 2>&1
find gpurun_out/tl -name "*kernel_trace.csv" -exec cp {} gpurun_out/timeline.csv \;
rm -rf gpurun_out/tl
python3 - "$N" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open("gpurun_out/timeline.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void gr::|^gr::|<.*|\(.*", "", r["Kernel_Name"])[:28]) for r in rows)
calls, cur = [], [ev[0]]
for p, q in zip(ev[:-1], ev[1:]):
    if q[0] - p[1] > 150e3: calls.append(cur); cur = []
    cur.append(q)
calls.append(cur)
calls = [c for c in calls if len(c) >= 40]
print("LM calls found:", len(calls), [len(c) for c in calls][-6:])
c = calls[-2] if len(calls) >= 2 else calls[-1]
print("call of %d kernels, span %.1f us, busy %.1f us" % (len(c), (c[-1][1] - c[0][0]) / 1e3, sum(e[1] - e[0] for e in c) / 1e3))
t0 = c[0][0]
prev = None
for e in c[:int(sys.argv[1])]:
    print("%9.2f  %-28s dur %7.2f  gap %6.2f" % ((e[0] - t0) / 1e3, e[2], (e[1] - e[0]) / 1e3, 0 if prev is None else (e[0] - prev) / 1e3))
    prev = e[1]
PY
rm -f gpurun_out/timeline.csv
