#!/bin/bash
# A/B of the heaviest-first launch order of conv_tile_kernel (EFG_TILE_LPT).  GPU box.
out=gpurun_out/lpt_sweep.txt; : > $out
for lvl in stem res2 res3 res4; do for kind in subm down; do for pass in fwd dgrad; do
  for x in 0 1; do
    EFG_TILE_LPT=$x python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass $pass 2>&1 | grep "^[rs][et]" | sed "s/^/lpt=$x  /" >> $out
  done
done; done; done
cat $out
