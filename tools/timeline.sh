#!/bin/bash
# kernel timeline of the default bench (on the GPU box): per-kernel gaps inside the timed LM loop
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --no-cpu-baseline --no-also "$@" > gpurun_out/timeline.log 2>&1
find gpurun_out/tl -name "*kernel_trace.csv" -exec cp {} gpurun_out/timeline.csv \;
rm -rf gpurun_out/tl
python3 - <<'PY'
import csv, re, collections
rows = list(csv.DictReader(open("gpurun_out/timeline.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void gr::|^gr::|<.*", "", r["Kernel_Name"])[:24]) for r in rows)
# LM steps = launches of the linearisation kernel; take the middle 30
chi = [i for i, e in enumerate(ev) if e[2] == "k_linearize"]
a, b = chi[len(chi)//2 - 15], chi[len(chi)//2 + 15]
seg = ev[a:b + 1]
wall = seg[-1][0] - seg[0][0]
busy = sum(e[1] - e[0] for e in seg[:-1])
print("steps 30 wall/step us %.1f busy/step us %.1f busy frac %.3f" % (wall / 30e3, busy / 30e3, busy / wall))
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for p, q in zip(seg[:-1], seg[1:]):
    gaps[p[2] + " -> " + q[2]].append(q[0] - p[1]); durs[p[2]].append(p[1] - p[0])
print("-- gaps (us): pair, count per step, mean")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print("  %-52s %5.2f %7.2f  total/step %6.2f" % (k, len(v) / 30, sum(v) / len(v) / 1e3, sum(v) / 30e3))
print("-- kernels (us): name, count per step, mean")
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    print("  %-28s %5.2f %7.2f  total/step %6.2f" % (k, len(v) / 30, sum(v) / len(v) / 1e3, sum(v) / 30e3))
PY
rm -f gpurun_out/timeline.csv
