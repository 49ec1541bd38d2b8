#!/bin/bash
# plan-walking weight gradient: A/B of its switches.  usage: scripts/wgt_ab.sh <tag> "ENV1=.. ENV2=.." "ENV..." ...
tag=$1; shift
out=gpurun_out/wgt_ab_$tag.txt; : > $out
for e in "$@"; do
  echo "== $e" >> $out
  for lvl in res2 res3 res4; do for kind in subm down; do
    env $e python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass wgrad 2>&1 | grep "^[rs][et]" | sed 's/m_in.*pairs.row//' >> $out
  done; done
done
cat $out
