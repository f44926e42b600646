# A/B of one SparseChol switch given as VAR=a,b (GPU box):  tools/r5_spchol_ab.sh GR_SPCHOL_MERGE_TRSM 0 1
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
env $VAR=$v timeout 300 python bench.py --solver dense-schur --no-cpu-baseline --no-also --pmc-traffic off --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$VAR=$v', l['value'], l['value_min'], l['value_max'], l['ms_per_step'], l.get('chi2_final')); print({k:(v['avg_us'],v['active_launches']) for k,v in l['roofline']['kernels'].items() if 'chol' in k})"
done
timeout 900 python -m pytest tests/test_gpu_cholesky.py -x -q -m gpu 2>&1 | tail -2
