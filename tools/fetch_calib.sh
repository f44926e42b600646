#!/bin/bash
# run ON THE GPU BOX: FETCH_SIZE per access pattern (tools/fetch_calib.hip) -> gpurun_out/fetch_calib.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/fetch_calib.hip -o build/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
echo "[" > $GRAFT_REPO_ROOT/gpurun_out/fetch_calib.json
first=1
for R in 0 8 16 24 48 64 192; do
  rm -rf /tmp/fc_$R
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fc_$R -o f -- $GRAFT_REPO_ROOT/build/fetch_calib $R 4096 8388608 > /tmp/fc_$R.out 2>&1
  python3 - $R >> $GRAFT_REPO_ROOT/gpurun_out/fetch_calib.json <<'PY'
import csv, glob, sys, re
R = sys.argv[1]
line = [l for l in open(f"/tmp/fc_{R}.out") if l.startswith("CALIB")][0].split()
asked = float(line[line.index("asked_bytes") + 1]); n = int(line[line.index("records") + 1])
val = None
for f in glob.glob(f"/tmp/fc_{R}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE" and ("k_gather" in row["Kernel_Name"] or "k_stream" in row["Kernel_Name"]):
            val = float(row["Counter_Value"])
import json
print(("" if R == "0" else ",") + json.dumps({"record_bytes": int(R), "pattern": "16 B per lane stream" if R == "0" else "one record per lane, random slots, each read once",
      "records": n, "asked_bytes": asked, "FETCH_SIZE_KB": val, "FETCH_SIZE_bytes_over_asked": None if val is None else round(val * 1024 / asked, 4),
      "bytes_per_record_by_counter": None if val is None or R == "0" else round(val * 1024 / n, 2)}))
PY
done
echo "]" >> $GRAFT_REPO_ROOT/gpurun_out/fetch_calib.json
cat $GRAFT_REPO_ROOT/gpurun_out/fetch_calib.json
