#!/bin/bash
# kernel-level A/B of the weight-gradient kernels on ONE box: scripts/wgt_prof_ab.sh <tag> "ENV.." "ENV.." ...
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$root
out=$root/gpurun_out/wgt_prof_ab_$tag.txt; : > $out
cd /tmp
for lvl in res2 res3 res4; do for kind in subm down; do
  echo "== $lvl $kind" >> $out
  for e in "$@"; do
    rm -rf /tmp/prof_wgt
    env $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wgt -o w -- python $root/scripts/ubench/conv_layer.py --level $lvl --kind $kind --pass wgrad > /tmp/w.log 2>&1
    python3 - "$e" >> $out <<PY
import csv,glob,sys
fs=glob.glob("/tmp/prof_wgt/**/*kernel_stats.csv", recursive=True)
t=[]
for r in csv.DictReader(open(fs[0])):
    n=r["Name"]
    if ("wgrad" in n or "wgt_reduce" in n) and int(r["Calls"])>5:
        t.append((n.replace("efg::(anonymous namespace)::","").split("(")[0][-28:], float(r["AverageNs"])/1e3))
print("   %-40s total %7.1f us   %s" % (sys.argv[1], sum(x[1] for x in t), "  ".join("%s %.1f" % x for x in t)))
PY
  done
done; done
cat $out
