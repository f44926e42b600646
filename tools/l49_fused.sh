cd $GRAFT_REPO_ROOT
for f in 0 1; do
  echo "GR_SCHUR_FUSED=$f"
  GR_SCHUR_FUSED=$f timeout 300 python bench.py --workload ladybug-49 --no-cpu-baseline --no-also --pmc-traffic off --repeats 5 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print(l['value'], l['value_min'], l['value_max'], l['ms_per_step'], l['pcg_iterations'], l['chi2_final'], l['accepted_steps']); print({k:(v['avg_us'],v['active_launches']) for k,v in l['roofline']['kernels'].items()})"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lm_paths.py -x -q -m gpu 2>&1 | tail -5
