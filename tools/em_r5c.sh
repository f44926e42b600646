# user-traits engine on the device-decided loop: parity tests, per-LM-iteration times (GPU box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_model.py -x -q -m gpu 2>&1 | tail -15
python - <<'PY'
from graphite_amd import synth
p = synth.make_config("ladybug-1723")
synth.write_bal("/tmp/l1723.txt", p)
PY
for cfg in "weighted stored" "weighted dynamic" "k3 stored" "pinhole stored"; do
  set -- $cfg
  echo "=== $1 $2 pcg (user-traits engine)"
  GRAPHITE_ENGINE=model GR_VERBOSE=1 timeout 300 ./build/test_engine_model /tmp/l1723.txt pcg 20 $1 $2 fp64 twice 2>&1 | grep -E "^ +(3|10|19) |SECOND|ENGINE_MODEL|LM:" | cut -c1-200
done > gpurun_out/em_time3.log 2>&1
cat gpurun_out/em_time3.log
