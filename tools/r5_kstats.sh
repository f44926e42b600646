# rocprofv3 --kernel-trace --stats of three bench.py commands at the end of round 5 (GPU box); csv summaries under gpurun_out/r05v5/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05v5; mkdir -p $O
run() { # tag, bench args
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o s -- python3 bench.py --no-cpu-baseline --no-also --pmc-traffic off --steps 20 --warmup 3 "$@" > $O/kstats_$tag.log 2>&1
  find /tmp/ks_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$tag.csv \;
  head -7 $O/kernel_stats_$tag.csv | cut -c1-140
}
run ladybug1723_f64_pcg
run ladybug49_f32_pcgschur --workload ladybug-49
run ladybug1723_f64_dense_schur --solver dense-schur
