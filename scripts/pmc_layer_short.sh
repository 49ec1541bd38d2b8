#!/bin/bash
# The three SQ counter sets of scripts/pmc_layer.sh only (wave / MFMA / issue accounting): pmc_layer_short.sh <tag> <conv_layer.py args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/pmc_layer
out=gpurun_out/pmc_layer/$tag.txt
python scripts/ubench/conv_layer.py "$@" 2>&1 | grep -v amdgpu > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pl$i; mkdir -p /tmp/pl$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pl$i -o r -- python scripts/ubench/conv_layer.py "$@" > /dev/null 2>&1
  python scripts/pmc_multi.py /tmp/pl$i/r_counter_collection.csv | grep -E "^kernel|conv_|tile" >> $out
done
cat $out
