# A/B of the lazy PCG direction (GR_PCG_LAZY) on the bench line: tools/ab_lazy.sh [bench args]
for g in 0 1 0 1; do
GR_PCG_LAZY=$g timeout 300 python bench.py --no-cpu-baseline --no-also --repeats 5 "$@" 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']; print('lazy=$g', l['value'], l['pcg_iterations'], l['chi2_final'], l['parity_rel'], {a:b['avg_us'] for a,b in k.items()})"
done
