cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -A4 "Name:.*VALUBusy\|Name:.*OccupancyPercent\|Name:.*MemUnitStalled\|Name:.*SQ_WAIT_ANY\b\|Name:.*SQ_ACTIVE_INST_ANY\|Name:.*SQ_WAVE_CYCLES\|Name:.*SQ_BUSY_CYCLES\|Name:.*MeanOccupancyPerCU\|Name:.*SQ_LEVEL_WAVES" | head -80
cd /tmp
for set in "VALUBusy" "OccupancyPercent" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rm -rf /tmp/pp_$tag
  rocprofv3 --pmc $set --output-format csv -d /tmp/pp_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --repeats 1 --steps 10 --warmup 2 --pmc-traffic off > /tmp/pp_$tag.log 2>&1
  echo "== $set rc=$?"
  python3 - /tmp/pp_$tag <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "k_pcg_operator" in k or "k_linearize<" in k:
        print(k, {c: round(sum(v) / len(v), 3) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
