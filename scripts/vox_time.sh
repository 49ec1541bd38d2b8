#!/bin/bash
# Per-launch time of the voxelizer's kernels (rocprofv3 kernel stats of scripts/bench_ops.py voxelize: 2 x 180k, 1 x 720k x 4
# sweeps, 8 x 180k).  usage: [ENV=...] scripts/vox_time.sh <tag> -> gpurun_out/vox_time_<tag>.txt
tag=${1:-base}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$root
rm -rf /tmp/prof_vox; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vox -o vox -- python $root/scripts/bench_ops.py voxelize > /tmp/vox.log 2>&1
cd $root
grep "hard_voxelize" /tmp/vox.log > gpurun_out/vox_time_$tag.txt
python3 - >> gpurun_out/vox_time_$tag.txt <<PY
import csv,glob
fs=glob.glob("/tmp/prof_vox/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if "vox" in r["Name"] or "fillBuffer" in r["Name"]:
        print("%-40s calls %4s avg %8.1f us  min %8.1f" % (r["Name"].replace("efg::(anonymous namespace)::","")[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
cat gpurun_out/vox_time_$tag.txt
