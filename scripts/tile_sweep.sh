#!/bin/bash
out=gpurun_out/wg_sweep.txt; : > $out
for lvl in res2 res3 res4; do for kind in subm down; do
  EFG_CONV_WG=0 python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass all 2>&1 | grep "^res" | sed 's/^/tiled      /' >> $out
  for shape in 81 41 42 22 24; do
    EFG_WG_MIN_TILES=0 EFG_WG_SHAPE=$shape python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass all 2>&1 | grep "^res" | sed "s/^/wg $shape      /" >> $out
  done
done; done
cat $out
