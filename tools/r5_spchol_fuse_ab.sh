cd $GRAFT_REPO_ROOT
for f in 1 2; do
GR_SPCHOL_FUSE=$f timeout 300 python bench.py --solver dense-schur --no-cpu-baseline --no-also --pmc-traffic off --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('fuse=$f', l['value'], l['value_min'], l['value_max'], l['ms_per_step']); print({k:(v['avg_us'],v['active_launches']) for k,v in l['roofline']['kernels'].items() if 'chol' in k})"
done
timeout 600 python -m pytest tests/test_gpu_cholesky.py -x -q -m gpu 2>&1 | tail -2
GR_SPCHOL_FUSE=2 timeout 600 python -m pytest tests/test_gpu_cholesky.py -x -q -m gpu 2>&1 | tail -2
