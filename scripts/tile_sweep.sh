#!/bin/bash
out=gpurun_out/xcd_sweep.txt; : > $out
for lvl in res2 res3 res4; do for kind in subm down; do
  for x in 1 0 2; do
    EFG_TILE_XCD=$x python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass fwd 2>&1 | grep "^res" | sed "s/^/xcd$x  /" >> $out
  done
done; done
cat $out
