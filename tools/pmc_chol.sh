#!/bin/bash
# MFMA-utilisation counters of the Cholesky GEMM kernels (on the GPU box): tools/pmc_chol.sh TAG n
TAG=$1; shift
export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_chol_$TAG -o p -- python3 tools/chol_bench.py "$@" > gpurun_out/pmc_chol_$TAG.log 2>&1
find gpurun_out/pmc_chol_$TAG -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_chol_$TAG.csv \;
rm -rf gpurun_out/pmc_chol_$TAG
python3 - gpurun_out/pmc_chol_$TAG.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# largest-grid dispatch of each chol gemm kernel
best = {}
for r in rows:
    n = r["Kernel_Name"]
    if "k_chol_gemm" not in n: continue
    key = n[:44]
    g = int(r["Grid_Size"])
    if key not in best or g > best[key][0]: best[key] = (g, r["Dispatch_Id"])
for key, (g, did) in best.items():
    vals = {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if r["Dispatch_Id"] == did}
    dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if r["Dispatch_Id"] == did][0]
    print(key, "grid", g, "dur_us", dur / 1e3, vals)
PY
# keep only the gemm rows (the full CSV is large)
python3 - gpurun_out/pmc_chol_$TAG.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_chol" in r["Kernel_Name"]]
keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
with open(sys.argv[1], "w", newline="") as fh:
    w = csv.DictWriter(fh, keep, quoting=csv.QUOTE_NONNUMERIC); w.writeheader()
    for r in rows:
        r = {k: r[k] for k in keep}; r["Kernel_Name"] = r["Kernel_Name"][:60]; w.writerow(r)
PY
