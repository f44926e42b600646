# user-traits engine after the closed-form test Jacobian and the K-wide tangents: parity tests, per-LM-iteration times, kernel stats (GPU box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_model.py -x -q -m gpu 2>&1 | tail -8
python - <<'PY'
from graphite_amd import synth
p = synth.make_config("ladybug-1723")
synth.write_bal("/tmp/l1723.txt", p)
PY
for cfg in "bal stored" "weighted stored" "weighted dynamic" "k3 stored" "pinhole stored"; do
  set -- $cfg
  echo "=== $1 $2 pcg (user-traits engine)"
  GRAPHITE_ENGINE=model timeout 300 ./build/test_engine_model /tmp/l1723.txt pcg 20 $1 $2 fp64 twice 2>&1 | grep -E "^ +(3|10|19) |SECOND|ENGINE_MODEL" | cut -c1-200
done > gpurun_out/em_time2.log 2>&1
cat gpurun_out/em_time2.log
cd /tmp && export TMPDIR=/tmp
for cfg in "weighted stored" "k3 stored"; do
  set -- $cfg
  GRAPHITE_ENGINE=model rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/em_prof_$1_$2 -o em -- $GRAFT_REPO_ROOT/build/test_engine_model /tmp/l1723.txt pcg 20 $1 $2 fp64 > /dev/null 2>&1
  f=$(find /tmp/em_prof_$1_$2 -name "*kernel_stats.csv" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/em2_kernel_stats_$1_$2.csv
  head -8 "$f" | cut -c1-150
done
