cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_box_fused_gpu.py tests/test_msda_gpu.py tests/test_determinism_gpu.py tests/test_model_full_golden.py tests/test_full_size_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-full-graph --no-arm --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['host_issue_ms_per_step'], d['kernels']['box_bwd_tile_kernel'], d['kernels']['box_bwd_kernel<32, false, 128>'])"
cat /proc/loadavg
