#!/bin/bash
# Issue / wait / pipe-busy counters of the sparse-conv kernels (GPU box): separate --pmc passes, kernel trace only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_conv
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_WAVES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/pc$i; mkdir -p /tmp/pc$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pc$i -o r -- python scripts/bench_ops.py spconv > /dev/null 2>&1
  python scripts/pmc_multi.py /tmp/pc$i/r_counter_collection.csv | grep -E "^kernel|conv_fwd|conv_wgrad" > gpurun_out/pmc_conv/set$i.csv
  cat gpurun_out/pmc_conv/set$i.csv
done
