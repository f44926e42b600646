# lineariser ablations INSIDE the solve (GR_DIAG build): tools/inloop_lin.sh WORKLOAD DTYPE
W=${1:-venice-1778}; D=${2:-f32}
for v in 0 1 2 4 8; do
GR_LIN_VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-also --repeats 1 --workload $W --dtype $D --solver pcg 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']; print('lin var $v (1 no record write, 2 no camera reduction, 4 no point gather, 8 no J math):', {a:b['avg_us'] for a,b in k.items() if 'linear' in a})"
done
