#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (KiB per launch) of the conv kernel of ONE layer: pmc_fetch_layer.sh <conv_layer.py args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pf; mkdir -p /tmp/pf
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pf -o r -- python scripts/ubench/conv_layer.py "$@" > /dev/null 2>&1
  python scripts/pmc_multi.py /tmp/pf/r_counter_collection.csv | grep -E "^kernel|conv_tile"
done
