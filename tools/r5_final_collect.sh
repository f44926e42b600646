# end-of-round-5 artifacts (GPU box): default bench line, dense-schur line + level trace, user-traits kernel stats / timeline / L49 timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v5; mkdir -p $O
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
python bench.py --solver dense-schur --no-cpu-baseline --no-also --steps 20 --warmup 3 > $O/bench_dense_schur.json 2>/dev/null
python bench.py --solver pcg-schur --no-cpu-baseline --no-also --steps 20 --warmup 3 > $O/bench_pcg_schur.json 2>/dev/null
bash tools/spchol_levels.sh > $O/spchol_levels.txt 2>&1
bash tools/em_timeline.sh weighted dynamic 30 > $O/em_timeline_weighted_recomputed.txt 2>&1
bash tools/em_timeline.sh weighted stored 30 > $O/em_timeline_weighted_stored.txt 2>&1
bash tools/em_l49.sh > $O/em_ladybug49_schur.txt 2>&1
bash tools/em_r5b.sh > $O/em_times.txt 2>&1
cp gpurun_out/em2_kernel_stats_weighted_stored.csv $O/ 2>/dev/null; cp gpurun_out/em2_kernel_stats_k3_stored.csv $O/ 2>/dev/null
./build/potrf_bench > $O/potrf_bench.txt 2>&1
tail -1 $O/bench_default.json | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print(l['value'], l['ms_per_step'], l.get('parity_rel'), l['roofline']['frac']); print([(a['workload'][:40], a['value'], a.get('parity_rel')) for a in l['also']])"
cat $O/bench_default.time | tail -3
