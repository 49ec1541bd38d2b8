#!/bin/bash
# L2-miss traffic (FETCH_SIZE, KiB) of the weight-gradient kernel per layer.  usage: scripts/wgt_traffic.sh <tag> "ENV" ...
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$root
out=$root/gpurun_out/wgt_traffic_$tag.txt; : > $out
cd /tmp
for lvl in res2 res4; do
  for e in "$@"; do
    rm -rf /tmp/pw; mkdir -p /tmp/pw
    env $e rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pw -o r -- python $root/scripts/ubench/conv_layer.py --level $lvl --kind subm --pass wgrad > /dev/null 2>&1
    echo "$lvl subm  $e: $(python $root/scripts/pmc_summary.py /tmp/pw/r_counter_collection.csv FETCH_SIZE | grep wgrad_tile)" >> $out
  done
done
cat $out
