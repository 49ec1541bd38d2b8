cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05c_gpu_tests.txt
cat gpurun_out/r05c_gpu_tests.txt
