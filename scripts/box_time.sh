#!/bin/bash
# Per-launch time of the fused box-attention kernels at the encoder / decoder shapes (rocprofv3 kernel stats of
# scripts/bench_ops.py box).  usage: [ENV=...] scripts/box_time.sh <tag> -> gpurun_out/box_time_<tag>.txt
tag=${1:-base}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$root
rm -rf /tmp/prof_box; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_box -o box -- python $root/scripts/bench_ops.py box > /tmp/box.log 2>&1
cd $root
grep "box fused" /tmp/box.log > gpurun_out/box_time_$tag.txt
python3 - >> gpurun_out/box_time_$tag.txt <<PY
import csv,glob
fs=glob.glob("/tmp/prof_box/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if "box" in r["Name"]:
        print("%-60s calls %4s avg %8.1f us" % (r["Name"].replace("efg::(anonymous namespace)::","")[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
cat gpurun_out/box_time_$tag.txt
