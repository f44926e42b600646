cd $GRAFT_REPO_ROOT
python - <<'PY'
from graphite_amd import synth
p = synth.make_config("ladybug-1723")
synth.write_bal("/tmp/l1723.txt", p)
PY
echo "=== reference bal.cu, unmodified, built-in model (verified hand-over)"
GR_VERBOSE=1 build/ref_examples/bal /tmp/l1723.txt --solver pcg --iterations 20 --verbose 2>&1 | grep -E "handed to|^ +1[0-9] |Optimization took|MSE" | tail -6
echo "=== reference bal.cu, unmodified, its own traits on the user-traits engine (GRAPHITE_ENGINE=model)"
GRAPHITE_ENGINE=model GR_VERBOSE=1 build/ref_examples/bal /tmp/l1723.txt --solver pcg --iterations 20 --verbose 2>&1 | grep -E "handed to|^ +1[0-9] |Optimization took|MSE" | tail -6
GRAPHITE_ENGINE=model GR_PROFILE_KERNELS=1 build/ref_examples/bal /tmp/l1723.txt --solver pcg --iterations 20 2>&1 | grep -E "^KERNEL|LM_LOOP" | awk '{printf "%s %s launches %s total_ms %s avg_us %.1f\n", $1, $2, $4, $8, 1000*$8/($6>0?$6:1)}'
echo "=== generic kernels"
GRAPHITE_GENERIC_ONLY=1 build/ref_examples/bal /tmp/l1723.txt --solver pcg --iterations 20 --verbose 2>&1 | grep -E "^ +1[0-9] |Optimization took|MSE" | tail -4
timeout 900 python -m pytest tests/test_engine_model.py -x -q -m gpu -k "masked or fixed_vertices" 2>&1 | tail -4
