#!/bin/bash
# kernel-level durations of the weight-gradient kernels per layer (rocprofv3 kernel stats).  usage: scripts/wgt_prof.sh <tag>
tag=${1:-base}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$root
out=$root/gpurun_out/wgt_prof_$tag.txt; : > $out
cd /tmp
for lvl in res2 res3 res4; do for kind in subm down; do
  rm -rf /tmp/prof_wgt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wgt -o w -- python $root/scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass wgrad > /tmp/w.log 2>&1
  echo "== $lvl $kind: $(grep '^[rs][et]' /tmp/w.log | sed 's/m_in.*pairs.row//')" >> $out
  python3 - >> $out <<PY
import csv,glob
fs=glob.glob("/tmp/prof_wgt/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if "wg" in r["Name"]:
        print("   %-50s calls %4s avg %8.1f us  min %8.1f" % (r["Name"].replace("efg::(anonymous namespace)::","")[:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done; done
cat $out
