cd $GRAFT_REPO_ROOT
python tools/fp32_stage_probe.py mini-50 2>&1 | tail -3
python tools/fp32_first_iteration_probe.py mini-50 2>&1 | grep "step:"
python tools/fp32_first_iteration_probe.py ladybug-49 2>&1 | grep "step:"
python tools/fp32_dx_probe.py 2>&1 | tail -9
timeout 900 python -m pytest tests/test_reference_examples.py -x -q -m gpu -k "low_precision" 2>&1 | grep -E "^E|passed|failed" | head -20
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lm_paths.py -x -q -m gpu 2>&1 | tail -3
