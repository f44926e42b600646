#!/bin/bash
# kernel timeline of the default bench line (on the GPU box): per LM call (= one timed repeat of bench.py) the span from its first to
# its last kernel, busy time, and the gaps / kernel times inside the calls
export TMPDIR=/tmp
# LM iterations per call = bench.py's --steps (its default: 20); forwarded arguments may override it
STEPS=20; prev=""
for a in "$@"; do
  if [ "$prev" = "--steps" ]; then STEPS="$a"; fi
  case "$a" in --steps=*) STEPS="${a#--steps=}";; esac
  prev="$a"
done
export GR_TL_STEPS="$STEPS"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --no-cpu-baseline --no-also "$@" > gpurun_out/timeline.log 2>&1
find gpurun_out/tl -name "*kernel_trace.csv" -exec cp {} gpurun_out/timeline.csv \;
rm -rf gpurun_out/tl
python3 - <<'PY'
import csv, re, collections, os
STEPS = int(os.environ.get("GR_TL_STEPS", "20"))
rows = list(csv.DictReader(open("gpurun_out/timeline.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void gr::|^gr::|<.*|\(.*", "", r["Kernel_Name"])[:24]) for r in rows)
# LM calls: runs of kernels separated by host gaps > 300 us that contain >= 15 k_linearize launches
calls, cur = [], [ev[0]]
for p, q in zip(ev[:-1], ev[1:]):
    if q[0] - p[1] > 300e3: calls.append(cur); cur = []
    cur.append(q)
calls.append(cur)
calls = [c for c in calls if sum(1 for e in c if e[2] == "k_linearize") >= max(1, STEPS * 3 // 4)]
print("LM calls found: %d (the last ones are the timed repeats + the profiled pass)" % len(calls))
for c in calls[-4:]:
    nlin = sum(1 for e in c if e[2] == "k_linearize")
    span = c[-1][1] - c[0][0]; busy = sum(e[1] - e[0] for e in c)
    print("  call: %d kernels, %d linearisations, span %.1f us = %.1f us per LM iteration (%d), busy %.1f us per iteration, idle %.1f us per iteration" % (len(c), nlin, span / 1e3, span / STEPS / 1e3, STEPS, busy / STEPS / 1e3, (span - busy) / STEPS / 1e3))
c = calls[-2] if len(calls) >= 2 else calls[-1]
nit = 1e3 * STEPS
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for p, q in zip(c[:-1], c[1:]):
    gaps[p[2] + " -> " + q[2]].append(q[0] - p[1])
for e in c: durs[e[2]].append(e[1] - e[0])
print("-- gaps inside one call (us): pair, count, mean, total per LM iteration")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("  %-52s %4d %7.2f  %6.2f" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / nit))
print("-- kernels of that call (us): name, count, mean, total per LM iteration")
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    print("  %-28s %4d %7.2f  %6.2f" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / nit))
PY
rm -f gpurun_out/timeline.csv
