cd $GRAFT_REPO_ROOT
bash tools/l1723_schur.sh
bash tools/l49_fused.sh
timeout 1500 python -m pytest tests/test_gpu_fullsize_oracle.py tests/test_gpu_fixed.py tests/test_gpu_random_sweep.py -x -q -m gpu 2>&1 | tail -5
