cd $GRAFT_REPO_ROOT
timeout 1200 bash scripts/round_profile.sh r05c > gpurun_out/r05c_round.log 2>&1
tail -3 gpurun_out/r05c_round.log
timeout 300 python scripts/ubench/soak.py 600 > gpurun_out/r05c_soak.txt 2>&1
tail -2 gpurun_out/r05c_soak.txt
