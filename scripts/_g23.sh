cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05d_gpu_tests.txt
cat gpurun_out/r05d_gpu_tests.txt
timeout 1200 bash scripts/round_profile.sh r05d > gpurun_out/r05d_round.log 2>&1
tail -3 gpurun_out/r05d_round.log
timeout 300 python scripts/ubench/soak.py 600 > gpurun_out/r05d_soak.txt 2>&1
tail -2 gpurun_out/r05d_soak.txt
