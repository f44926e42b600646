# user-traits engine on the Ladybug-1723 shape: per-LM-iteration times + rocprofv3 kernel stats (run on the GPU box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY'
from graphite_amd import synth
p = synth.make_config("ladybug-1723")
synth.write_bal("/tmp/l1723.txt", p)
PY
for mode in bal weighted k3; do
  for jac in stored dynamic; do
    if [ $mode = k3 ] && [ $jac = dynamic ]; then continue; fi
    echo "=== $mode $jac pcg (user-traits engine)"
    GRAPHITE_ENGINE=model GR_VERBOSE=1 timeout 300 ./build/test_engine_model /tmp/l1723.txt pcg 20 $mode $jac fp64 twice 2>&1 | grep -E "^ +[0-9]+ |SECOND|TOTAL|ENGINE_MODEL|operator|LM:" | tail -30
  done
done > gpurun_out/em_time.log 2>&1
echo "=== bal built-in" >> gpurun_out/em_time.log
timeout 300 ./build/test_engine_model /tmp/l1723.txt pcg 20 bal stored fp64 twice 2>&1 | grep -E "^ +[0-9]+ |SECOND|TOTAL|ENGINE_MODEL" | tail -8 >> gpurun_out/em_time.log
echo "=== weighted generic" >> gpurun_out/em_time.log
GRAPHITE_GENERIC_ONLY=1 timeout 300 ./build/test_engine_model /tmp/l1723.txt pcg 20 weighted 2>&1 | grep -E "^ +[0-9]+ |TOTAL" | tail -6 >> gpurun_out/em_time.log
cd /tmp && export TMPDIR=/tmp
GRAPHITE_ENGINE=model rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/em_prof -o em -- $GRAFT_REPO_ROOT/build/test_engine_model /tmp/l1723.txt pcg 20 weighted stored fp64 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/em_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/em_kernel_stats_weighted_stored.csv
rm -rf gpurun_out/em_prof
cat gpurun_out/em_time.log | cut -c1-170
head -20 gpurun_out/em_kernel_stats_weighted_stored.csv | cut -c1-200
