cd $GRAFT_REPO_ROOT
bash tools/l49_fused.sh 2>&1 | head -8
bash tools/timeline.sh --workload ladybug-49 --pmc-traffic off --repeats 3 > gpurun_out/timeline_l49.txt 2>&1
bash tools/timeline_dump.sh 40 --workload ladybug-49 --pmc-traffic off --repeats 3 > gpurun_out/timeline_dump_l49.txt 2>&1
tail -30 gpurun_out/timeline_l49.txt
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_pytest.log 2>&1; tail -15 gpurun_out/full_gpu_pytest.log
