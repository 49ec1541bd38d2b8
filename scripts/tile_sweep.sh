#!/bin/bash
out=gpurun_out/wgrad_sweep.txt; : > $out
for lvl in res2 res3 res4; do for kind in subm down; do
  for fill in 512 1024 2048 4096; do
    EFG_WGRAD_FILL=$fill python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass wgrad 2>&1 | grep "^res" | sed "s/^/fill$fill  /" >> $out
  done
done; done
cat $out
python scripts/bench_ops.py spconv --detail > gpurun_out/spconv_detail_r02b.txt 2>&1; tail -12 gpurun_out/spconv_detail_r02b.txt
