# end-of-round-6 artifacts (GPU box): default bench line, rocprofv3 kernel stats of the headline / Ladybug-49 / Venice commands,
# resident-launch A/B; everything under gpurun_out/r06/ (copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06; mkdir -p $O
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
run() { # tag, bench args
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o s -- python3 bench.py --no-cpu-baseline --no-also --pmc-traffic off --steps 20 --warmup 3 "$@" > $O/kstats_$tag.log 2>&1
  find /tmp/ks_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$tag.csv \;
  head -7 $O/kernel_stats_$tag.csv | cut -c1-140
}
run ladybug1723_f64_pcg
run ladybug49_f32_pcgschur --workload ladybug-49
run venice1778_f32_pcg --workload venice-1778 --dtype f32
GR_PCG_RESIDENT=1 run ladybug1723_f64_pcg_resident
python tools/rp_ab.py ladybug-1723 > $O/resident_ab.txt 2>&1
tail -1 $O/bench_default.json | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print(l['value'], l['ms_per_step'], l.get('parity_rel'), l['roofline']['frac']); print([(a['workload'][:40], a['value'], a.get('parity_rel')) for a in l['also']])"
cat $O/bench_default.time | tail -3
