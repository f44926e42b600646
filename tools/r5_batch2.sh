cd $GRAFT_REPO_ROOT
bash tools/fetch_calib.sh 2>&1 | tail -12
bash tools/pmc_probe.sh 2>&1 | tail -40
python tools/fp32_stage_probe.py mini-50 2>&1 | tail -12
python tools/fp32_dx_probe.py 2>&1 | tail -15
