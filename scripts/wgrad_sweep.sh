#!/bin/bash
# Weight gradient of every covered sparse-conv layer shape of the res18 backbone (2 x 180k-point scenes), timed alone:
# table kernel (EFG_WGRAD_TILED=0) against the plan-walking kernel.  usage: [ENV=...] scripts/wgrad_sweep.sh <tag>
tag=${1:-base}
out=gpurun_out/wgrad_sweep_$tag.txt; : > $out
for t in 0 1; do
  echo "EFG_WGRAD_TILED=$t" >> $out
  for lvl in res2 res3 res4; do for kind in subm down; do
    EFG_WGRAD_TILED=$t python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass wgrad 2>&1 | grep "^[rs][et]" >> $out
  done; done
done
cat $out
