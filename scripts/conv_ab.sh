#!/bin/bash
# Forward / dgrad of every 64+-channel sparse-conv layer shape, A/B of environment switches on ONE box (event timings of
# scripts/ubench/conv_layer.py).  usage: scripts/conv_ab.sh <tag> "ENV.." "ENV.." ...
tag=$1; shift
out=gpurun_out/conv_ab_$tag.txt; : > $out
for lvl in res2 res3 res4; do for kind in subm down; do
  echo "== $lvl $kind" >> $out
  for e in "$@"; do
    r=$(env $e python scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass all 2>&1 | grep "^[rs][et]" | sed 's/.*pairs.row [0-9.]*//')
    printf "   %-44s %s\n" "$e" "$r" >> $out
  done
done; done
cat $out
