cd $GRAFT_REPO_ROOT
python tools/fp32_dx_probe.py 2>&1 | tail -9
python tools/fp32_parity_probe.py 2>&1 | tail -12
timeout 900 python -m pytest tests/test_reference_examples.py -x -q -m gpu -k "low_precision" 2>&1 | grep -E "^E|passed|failed" | head -20
