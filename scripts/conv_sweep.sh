#!/bin/bash
# Every sparse-conv layer shape of the res18 backbone (2 x 180k-point scenes), fwd / dgrad / wgrad timed alone.
# usage: [ENV=...] scripts/conv_sweep.sh <tag>   -> gpurun_out/conv_sweep_<tag>.txt
tag=${1:-base}
out=gpurun_out/conv_sweep_$tag.txt; : > $out
for lvl in stem res2 res3 res4; do for kind in subm down; do
  python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass all 2>&1 | grep "^[rs][et]" >> $out
done; done
cat $out
