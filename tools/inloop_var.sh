for v in 0 1 2 3; do
GR_OP_VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-also --repeats 1 --workload venice-1778 --dtype f32 --solver pcg --pcg-tol 0 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']; print('var $v', l['value'], l['pcg_iterations'], {a:(b['avg_us'],b['active_launches']) for a,b in k.items()})"
done
