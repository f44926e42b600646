# in-loop kernel times of one workload for several point-tile counts (GR_PTILES), 10 fixed inner iterations
W=${1:-venice-1778}; D=${2:-f32}; shift; shift
for k in "$@"; do
GR_PTILES=$k timeout 300 python bench.py --no-cpu-baseline --no-also --repeats 1 --workload $W --dtype $D --solver pcg --pcg-tol 0 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']; print('K=$k', l['value'], l['pcg_iterations'], {a:b['avg_us'] for a,b in k.items()})"
done
