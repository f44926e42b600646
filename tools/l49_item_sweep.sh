cd $GRAFT_REPO_ROOT
for it in 14 28 42 56 84; do
  echo "GR_SCHUR_ITEM=$it"
  GR_SCHUR_ITEM=$it python bench.py --workload ladybug-49 --no-cpu-baseline --no-also --pmc-traffic off --repeats 5 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print(l['value'], l['value_min'], l['value_max'], l['ms_per_step']); print({k:v['avg_us'] for k,v in l['roofline']['kernels'].items()})"
done
