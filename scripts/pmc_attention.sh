#!/bin/bash
# Issue-level counters of the attention kernels (scripts/ubench/attention_bench.py): usage pmc_attention.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/pmc_attention
out=gpurun_out/pmc_attention/$tag.txt
python scripts/ubench/attention_bench.py "$@" 2>&1 | grep -v amdgpu | tail -2 > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/pa$i; mkdir -p /tmp/pa$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pa$i -o r -- python scripts/ubench/attention_bench.py "$@" > /dev/null 2>&1
  python scripts/pmc_multi.py /tmp/pa$i/r_counter_collection.csv | grep -E "^kernel|attn_" >> $out
done
cat $out
