#!/bin/bash
out=gpurun_out/gen_ab.txt; : > $out
for lvl in stem res2 res3 res4; do for kind in subm down; do for pass in fwd dgrad; do
  for x in 1 0; do
    EFG_CONV_TILED=$x python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass $pass 2>&1 | grep "^[rs][et]" | sed "s/^/tiled=$x  /" >> $out
  done
done; done; done
cat $out
