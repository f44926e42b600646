cd $GRAFT_REPO_ROOT
for f in 1 2; do
GR_SCHUR_FUSED=$f timeout 300 python bench.py --workload ladybug-49 --no-cpu-baseline --no-also --pmc-traffic off --repeats 5 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('L49 schur_fused=$f', l['value'], l['value_min'], l['value_max'], l['ms_per_step'], l['pcg_iterations'], l['accepted_steps'], l['chi2_final'], l.get('parity_rel')); print({k:(v['avg_us'],v['active_launches']) for k,v in l['roofline']['kernels'].items()})"
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lm_paths.py tests/test_gpu_random_sweep.py tests/test_gpu_fixed.py -x -q -m gpu 2>&1 | tail -3
