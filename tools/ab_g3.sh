# A/B of the g3 layout (GR_G3_GATHER) inside the solve: tools/ab_g3.sh WORKLOAD DTYPE
W=${1:-ladybug-1723}; D=${2:-f64}
for g in 0 1 0 1; do
GR_G3_GATHER=$g timeout 300 python bench.py --no-cpu-baseline --no-also --repeats 5 --workload $W --dtype $D --solver pcg --pcg-tol 0 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']; print('gather=$g', l['value'], l['pcg_iterations'], l['chi2_final'], l['parity_rel'], {a:b['avg_us'] for a,b in k.items()})"
done
