#!/bin/bash
# kernel-by-kernel trace of the LAST LM iterations of the user-traits engine test client (second optimiser call, cached problem)
#   tools/em_timeline.sh <mode> <stored|dynamic> [N kernels]        (on the GPU box)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MODE=${1:-weighted}; JAC=${2:-dynamic}; N=${3:-60}
python3 - <<'PY'
import os
from graphite_amd import synth
if not os.path.exists("/tmp/l1723.txt"):
    synth.write_bal("/tmp/l1723.txt", synth.make_config("ladybug-1723"))
PY
mkdir -p gpurun_out
cd /tmp
GRAPHITE_ENGINE=model rocprofv3 --kernel-trace --output-format csv -d /tmp/emtl -o t -- $GRAFT_REPO_ROOT/build/test_engine_model /tmp/l1723.txt pcg 20 $MODE $JAC fp64 twice > /tmp/emtl.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/emtl -name "*kernel_trace.csv" -exec cp {} gpurun_out/em_timeline.csv \;
python3 - "$N" <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open("gpurun_out/em_timeline.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void (gr|graphite::detail)::|^gr::|<.*|\(.*", "", r["Kernel_Name"])[:28]) for r in rows)
n = int(sys.argv[1])
# the last LM call: from the 21st k_em_linearize from the end
lin = [i for i, e in enumerate(ev) if e[2] == "k_em_linearize"]
start = lin[-20] if len(lin) >= 20 else 0
c = ev[start:]
span = c[-1][1] - c[0][0]; busy = sum(e[1] - e[0] for e in c)
print("last 20 linearisations: %d kernels, span %.1f us = %.1f per LM iteration, busy %.1f per iteration" % (len(c), span / 1e3, span / 20e3, busy / 20e3))
durs = collections.defaultdict(list); gaps = collections.defaultdict(list)
for e in c: durs[e[2]].append(e[1] - e[0])
for p, q in zip(c[:-1], c[1:]): gaps[p[2] + " -> " + q[2]].append(q[0] - p[1])
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])): print("  %-28s %4d  mean %7.2f  per LM iteration %7.2f" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 20e3))
print("-- gaps")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:10]: print("  %-58s %4d  mean %7.2f  per LM iteration %7.2f" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 20e3))
t0 = c[0][0]; prev = None
k0 = max(0, len(c) - n)
for e in c[k0:]:
    print("%9.2f  %-28s dur %7.2f  gap %6.2f" % ((e[0] - t0) / 1e3, e[2], (e[1] - e[0]) / 1e3, 0 if prev is None else (e[0] - prev) / 1e3))
    prev = e[1]
PY
rm -f gpurun_out/em_timeline.csv
