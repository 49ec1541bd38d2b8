#!/bin/bash
# HBM traffic of our kernels from hardware counters (GPU box).  Separate --pmc passes, no tracing
# domains besides --kernel-trace (MI355X_MICROARCH.md §HBM / gpurun rules).  Output: gpurun_out/<tag>/
tag=${1:-pmc}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/$tag.$ctr && mkdir -p /tmp/$tag.$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/$tag.$ctr -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/$tag/bench.$ctr.log 2>&1
  ls /tmp/$tag.$ctr | head
  python scripts/pmc_summary.py /tmp/$tag.$ctr/r_counter_collection.csv $ctr > gpurun_out/$tag/$ctr.summary.csv
  head -5 gpurun_out/$tag/$ctr.summary.csv
done
