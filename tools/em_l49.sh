# user-traits engine on a Ladybug-49-sized graph (the local-BA size of a SLAM back end), Schur PCG: per-LM-iteration time (GPU box)
cd $GRAFT_REPO_ROOT
python - <<'PY'
from graphite_amd import synth
synth.write_bal("/tmp/l49.txt", synth.make_config("ladybug-49"))
PY
for cfg in "weighted stored fp64" "weighted stored fp32" "pinhole stored fp64" "bal stored fp64"; do
  set -- $cfg
  for f in 2 1 0; do
    echo "=== $1 $2 $3 pcg-schur GR_SCHUR_FUSED=$f"
    GR_SCHUR_FUSED=$f GRAPHITE_ENGINE=model GR_VERBOSE=1 timeout 120 ./build/test_engine_model /tmp/l49.txt pcg-schur 20 $1 $2 $3 twice 2>&1 | grep -E "^ +(19) |SECOND|LM:" | cut -c1-220
  done
done
echo "=== built-in model"; timeout 120 ./build/test_engine_model /tmp/l49.txt pcg-schur 20 bal stored fp64 twice 2>&1 | grep -E "^ +(19) |SECOND" | cut -c1-200
