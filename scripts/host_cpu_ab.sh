#!/bin/bash
# Host CPU per training process under a few HIP / ROCr runtime settings (scripts/ubench/host_threads.py).  GPU box.
out=gpurun_out/host_cpu_ab.txt; : > $out
run() { echo "== $*" >> $out; env OMP_NUM_THREADS=1 "$@" python scripts/ubench/host_threads.py 2>&1 | grep -E "ms/step|thread |sum of" >> $out; }
run A=0
run AMD_DIRECT_DISPATCH=0
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=200
run ROC_CPU_WAIT_FOR_SIGNAL=0
run GPU_MAX_HW_QUEUES=4
run ROC_SIGNAL_POOL_SIZE=4096
cat $out
