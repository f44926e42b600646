#!/bin/bash
# Run ON THE GPU BOX: SQ activity counters of the per-observation kernels for one bench.py configuration (two passes).
#   tools/pmc_sq.sh TAG [bench.py args...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
ARGS="--no-cpu-baseline --no-also --repeats 1 --steps 10 --warmup 2 $*"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/a -o a -- python3 bench.py $ARGS > $OUT/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- python3 bench.py $ARGS > $OUT/b.log 2>&1
python3 - "$OUT" <<'PY'
import csv, sys, os, collections, glob
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*counter_collection.csv")) + glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k, d in sorted(agg.items()):
        if not any(x in k for x in ("k_linearize", "k_pcg", "k_block", "k_apply", "k_schur", "k_finalize", "k_bschur", "k_backsub")): continue
        line = k.ljust(62) + " ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(d.items()))
        print(line); fh.write(line + "\n")
PY
