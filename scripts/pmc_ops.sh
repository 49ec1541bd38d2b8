#!/bin/bash
# SQ counter sets (wave / issue / LDS accounting) of the kernels of one scripts/bench_ops.py target:
#   scripts/pmc_ops.sh <tag> <bench_ops target> <kernel-name regex>      e.g.  pmc_ops.sh quad box box_
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
tag=${1:-base}; target=${2:-box}; pat=${3:-box_}
out=gpurun_out/pmc_${target}_$tag.txt; : > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1)); rm -rf /tmp/pb$i; mkdir -p /tmp/pb$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pb$i -o r -- python scripts/bench_ops.py $target > /dev/null 2>&1
  python scripts/pmc_multi.py /tmp/pb$i/r_counter_collection.csv | grep -E "^kernel|$pat" >> $out
done
cat $out
