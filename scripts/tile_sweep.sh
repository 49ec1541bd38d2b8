#!/bin/bash
# A/B of the 16-byte-gather tile kernel (EFG_TILE_V4=1) against the 4-byte path, per backbone layer.  GPU box.
out=gpurun_out/v4_sweep.txt; : > $out
for lvl in res2 res3 res4; do for kind in subm down; do for pass in fwd dgrad; do
  for x in 0 1; do
    EFG_TILE_V4=$x python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass $pass 2>&1 | grep "^res" | sed "s/^/v4=$x  /" >> $out
  done
done; done; done
cat $out
