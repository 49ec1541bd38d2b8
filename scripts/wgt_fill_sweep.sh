#!/bin/bash
# plan-walking weight gradient: workgroup count sweep (EFG_WGT_FILL).  usage: scripts/wgt_fill_sweep.sh <tag> [fills...]
tag=${1:-base}; shift
out=gpurun_out/wgt_fill_$tag.txt; : > $out
for f in ${@:-512 1024 2048 4096}; do
  echo "EFG_WGT_FILL=$f" >> $out
  for lvl in res2 res3 res4; do for kind in subm down; do
    EFG_WGT_FILL=$f python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass wgrad 2>&1 | grep "^[rs][et]" | sed 's/m_in.*pairs.row//' >> $out
  done; done
done
cat $out
