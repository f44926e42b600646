set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY'
from graphite_amd import synth
p = synth.make_config("mini-50")
synth.write_bal("/tmp/mini50.txt", p)
print(p.shape)
PY
for mode in bal weighted k3 pinhole; do
  for solver in pcg pcg-schur; do
    echo "=== $mode $solver engine(model)"
    GRAPHITE_ENGINE=model GR_VERBOSE=1 timeout 120 ./build/test_engine_model /tmp/mini50.txt $solver 6 $mode 2>&1 | grep -v "^\[graphite\] hand-over:" | tail -22
    echo "=== $mode $solver generic"
    GRAPHITE_GENERIC_ONLY=1 timeout 120 ./build/test_engine_model /tmp/mini50.txt $solver 6 $mode 2>&1 | tail -14
  done
done > gpurun_out/em_smoke.log 2>&1
tail -5 gpurun_out/em_smoke.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
