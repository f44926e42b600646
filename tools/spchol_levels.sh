#!/bin/bash
# ON THE GPU BOX: per-launch durations and gaps of the nested-dissection tile Cholesky inside one LM iteration of the direct Schur
# solver (bench.py --solver dense-schur): where the level chain's time goes.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/spchol; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- python3 bench.py --solver dense-schur --no-cpu-baseline --no-also --steps 4 --warmup 2 --repeats 1 "$@" > $OUT/run.log 2>&1
find $OUT/tl -name "*kernel_trace.csv" -exec cp {} $OUT/trace.csv \;
rm -rf $OUT/tl
python3 - "$OUT/trace.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void gr::|^gr::|\(.*", "", r["Kernel_Name"])[:40], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)) for r in rows)
# the last factorisation = from the last k_sp_clear to the last k_sp_unpermute
last_clear = max(i for i, e in enumerate(ev) if e[2].startswith("k_sp_clear"))
end = max(i for i, e in enumerate(ev) if e[2].startswith("k_sp_unpermute"))
seg = ev[last_clear:end + 1]
t0 = seg[0][0]
print("%-42s %8s %9s %9s %8s" % ("kernel", "wgs", "start us", "dur us", "gap us"))
prev_end = None
tot = {}
for s, e, n, g in seg:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-42s %8d %9.1f %9.1f %8.1f" % (n, g // 256 if g else 0, (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev_end = max(prev_end or e, e)
    tot[n] = tot.get(n, 0) + (e - s) / 1e3
print("span %.1f us" % ((seg[-1][1] - t0) / 1e3))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]): print("  %-42s %9.1f us" % (n, v))
PY
