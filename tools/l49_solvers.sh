cd $GRAFT_REPO_ROOT
for s in pcg-schur pcg-schur-implicit pcg; do
  echo "solver=$s"
  python bench.py --workload ladybug-49 --solver $s --no-cpu-baseline --no-also --pmc-traffic off --repeats 5 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print(l['value'], l['value_min'], l['value_max'], l['ms_per_step'], l['pcg_iterations'], l['chi2_final']); print({k:(v['avg_us'],v['active_launches']) for k,v in l['roofline']['kernels'].items()})"
done
