#!/bin/bash
# cgroup CPU throttling of the training process vs the host thread-pool size (GPU box).
thr() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
run() { python bench.py --no-cpu-baseline --steps 40 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for cfg in "EFG_HOST_THREADS=256 EFG_MANUAL_GC=0" "EFG_HOST_THREADS=4 EFG_MANUAL_GC=0" "EFG_HOST_THREADS=4 EFG_MANUAL_GC=1" "EFG_HOST_THREADS=256 EFG_MANUAL_GC=1" "EFG_HOST_THREADS=4 EFG_MANUAL_GC=1" "EFG_HOST_THREADS=4 EFG_MANUAL_GC=1 EFG_FUSED_LINEAR=0"; do
  echo -n "$cfg | before: $(thr) | "
  env $cfg bash -c "$(declare -f run); run"
  echo "      after: $(thr)"
done
