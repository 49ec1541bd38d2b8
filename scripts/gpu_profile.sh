#!/bin/bash
# usage: scripts/gpu_profile.sh <tag> [bench args...]   (run on the GPU box via gpurun)
# Writes the rocprofv3 kernel stats CSV (small) under gpurun_out/<tag>/; the raw trace stays in /tmp.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/$tag gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag -o r -- python bench.py "$@" > gpurun_out/$tag/bench.log 2>&1
cp /tmp/$tag/r_kernel_stats.csv /tmp/$tag/r_domain_stats.csv gpurun_out/$tag/ 2>/dev/null
tail -1 gpurun_out/$tag/bench.log | cut -c1-400
