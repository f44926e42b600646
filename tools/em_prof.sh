cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY'
from graphite_amd import synth
p = synth.make_config("ladybug-1723")
synth.write_bal("/tmp/l1723.txt", p)
PY
cd /tmp && export TMPDIR=/tmp
for cfg in "weighted stored" "weighted dynamic" "k3 stored"; do
  set -- $cfg
  GRAPHITE_ENGINE=model rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/em_prof_$1_$2 -o em -- $GRAFT_REPO_ROOT/build/test_engine_model /tmp/l1723.txt pcg 20 $1 $2 fp64 > /dev/null 2>&1
  f=$(find /tmp/em_prof_$1_$2 -name "*kernel_stats.csv" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/em_kernel_stats_$1_$2.csv
done
cd $GRAFT_REPO_ROOT
head -16 gpurun_out/em_kernel_stats_weighted_stored.csv | cut -c1-220
