cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload ladybug-49 --no-cpu-baseline --no-also --pmc-traffic off --repeats 5 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('L49', l['value'], l['value_min'], l['value_max'], l['ms_per_step'], l['pcg_iterations'], l['chi2_final']); print({k:(v['avg_us'],v['active_launches']) for k,v in l['roofline']['kernels'].items()})"
timeout 300 python bench.py --solver pcg-schur --no-cpu-baseline --no-also --pmc-traffic off --repeats 5 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('L1723 pcg-schur', l['value'], l['value_min'], l['value_max'], l['ms_per_step'], l['pcg_iterations'], l['chi2_final']); print({k:(v['avg_us'],v['active_launches']) for k,v in l['roofline']['kernels'].items()})"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lm_paths.py tests/test_gpu_random_sweep.py -x -q -m gpu -k "schur or Schur" 2>&1 | tail -3
