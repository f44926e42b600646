for n in 150 300 600 1000; do
python - $n <<'PY'
import sys; sys.path.insert(0, '.')
from graphite_amd import synth
n = int(sys.argv[1])
p0, fx, e, m, info, tr = synth.make_pose_graph(n)
synth.write_pose_graph('/tmp/g.txt', p0, fx, e, m, info, huber_delta=0.0)
print("POSES", n, "columns", 3*(n-1))
PY
for mode in sparse dense; do
  if [ $mode = dense ]; then export GRAPHITE_LDLT_SPARSE_MIN=100000000; else export GRAPHITE_LDLT_SPARSE_MIN=1; fi
  build/test_pose_graph /tmp/g.txt eigen 5 manual 10 1.0 2>&1 | awk -v m=$mode 'NF==6 && $1 ~ /^[0-9]+$/ {if ($1>0) {s+=$5; k++}} /SPARSE_FACT/{sp=$2} END{printf "  %s (SPARSE_FACTORISATION %s): %.3f ms per LM iteration\n", m, sp, 1e3*s/k}'
done
done
