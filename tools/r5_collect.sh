cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh l1723 --no-also --pmc-traffic off > /dev/null 2>&1
bash tools/collect_profiles.sh l49 --workload ladybug-49 --no-also --pmc-traffic off > /dev/null 2>&1
bash tools/collect_profiles.sh l1723schur --workload ladybug-1723 --solver pcg-schur --no-also --pmc-traffic off --steps 10 > /dev/null 2>&1
ls gpurun_out/prof_l1723 gpurun_out/prof_l49 gpurun_out/prof_l1723schur
head -8 gpurun_out/prof_l1723/kernel_stats.csv | cut -c1-160
bash tools/timeline.sh --pmc-traffic off > gpurun_out/timeline_l1723.txt 2>&1; tail -28 gpurun_out/timeline_l1723.txt
