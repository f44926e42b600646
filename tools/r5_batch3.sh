cd $GRAFT_REPO_ROOT
python tools/fp32_first_iteration_probe.py mini-50 2>&1 | tail -9
python tools/fp32_first_iteration_probe.py ladybug-49 2>&1 | tail -9
timeout 900 python -m pytest tests/test_generic_api.py tests/test_reference_examples.py -x -q -m gpu 2>&1 | tail -5
