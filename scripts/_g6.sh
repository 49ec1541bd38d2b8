cd $GRAFT_REPO_ROOT
python scripts/ubench/launch_census.py > gpurun_out/r05c_launch_census.txt 2>&1
tail -5 gpurun_out/r05c_launch_census.txt
