# the three forms of the matrix-free PCG on the headline line (direction kernel / lazy direction / single reduction): GPU box
cd $GRAFT_REPO_ROOT
for cfg in "GR_PCG_LAZY=0 GR_PCG_CG=0" "GR_PCG_LAZY=1 GR_PCG_CG=0" "GR_PCG_LAZY=0 GR_PCG_CG=1"; do
env $cfg timeout 300 python bench.py --no-cpu-baseline --no-also --pmc-traffic off --repeats 5 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$cfg', l['value'], l['value_min'], l['value_max'], l['ms_per_step'], l['pcg_iterations'], l.get('parity_rel'))"
done
