#!/bin/bash
# Knock-out A/B of conv_fat_kernel (diagnostic builds of scripts/build_ab.sh with -DEFG_FKNOCK=<bits>): which part of a
# step the launch time belongs to.  Results of the knocked-out builds are WRONG by construction; timings only.
out=gpurun_out/conv_fknock.txt; : > $out
for waves in 1 2; do for lvl in res2 res3 res4; do
  echo "== $lvl subm, $waves wave(s) per SIMD" >> $out
  for k in 0 1 2 3 4 7 8 15; do
    if [ $k = 0 ]; then e="EFG_X=0"; else e="EFG_HIP_LIB_AB=libefg_fk$k.so"; fi
    r=$(env $e EFG_FAT_WAVES=$waves python scripts/ubench/conv_layer.py --level $lvl --kind subm --pass fwd 2>&1 | grep "^[rs][et]" | sed 's/.*pairs.row [0-9.]*//')
    printf "   knock %-3s %s\n" "$k" "$r" >> $out
  done
done; done
cat $out
