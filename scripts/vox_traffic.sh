#!/bin/bash
# HBM-side traffic of one hard_voxelize call (all its kernels + the memset), both implementations:
#   scripts/vox_traffic.sh <tag> [points sweeps scenes]   -> gpurun_out/vox_traffic_<tag>.txt
# Two rocprofv3 --pmc passes per implementation (FETCH_SIZE and WRITE_SIZE do not fit one pass); bytes per call =
# (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over every kernel of the process / calls (MI355X_MICROARCH.md, HBM: on gfx950
# FETCH_SIZE reports half the bytes of a wide streaming read; WRITE_SIZE taken as is).
tag=${1:-base}; n=${2:-180000}; sw=${3:-1}; nb=${4:-2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$root
out=$root/gpurun_out/vox_traffic_$tag.txt; : > $out
for impl in hash bins; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/vt; mkdir -p /tmp/vt; cd /tmp
    EFG_VOX_IMPL=$impl rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/vt -o r -- python $root/scripts/vox_one.py $n $sw $nb 10 > /tmp/vt/log.txt 2>&1
    cd $root
    python3 - $impl $ctr >> $out <<'PY'
import csv, glob, re, sys
impl, ctr = sys.argv[1], sys.argv[2]
log = open("/tmp/vt/log.txt").read()
m = re.search(r"VOX calls (\d+) voxels (\d+) alg_bytes (\d+) us_per_call (\S+)", log)
calls, vox, alg = int(m.group(1)), int(m.group(2)), int(m.group(3))
tot, per = 0.0, {}
for fn in glob.glob("/tmp/vt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != ctr:
            continue
        k = r["Kernel_Name"]
        if not ("vox_" in k or "fillBuffer" in k):
            continue
        k = re.sub(r"\(.*", "", k.replace("efg::(anonymous namespace)::", "").replace("void ", ""))[:40]
        per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
        tot += float(r["Counter_Value"])
print("%s %s per_call_KB %.1f  (calls %d, voxels %d, alg_bytes %d)  " % (impl, ctr, tot / calls, calls, vox, alg) +
      " ".join("%s=%.0f" % (k, v / calls) for k, v in sorted(per.items(), key=lambda kv: -kv[1])))
PY
  done
done
python3 - $out <<'PY'
import re, sys
rows = {}
for l in open(sys.argv[1]):
    m = re.match(r"(\w+) (\w+) per_call_KB (\S+)  \(calls \d+, voxels \d+, alg_bytes (\d+)", l)
    if m:
        rows[(m.group(1), m.group(2))] = float(m.group(3)); alg = int(m.group(4))
with open(sys.argv[1], "a") as f:
    for impl in ("hash", "bins"):
        b = (2 * rows[(impl, "FETCH_SIZE")] + rows[(impl, "WRITE_SIZE")]) * 1024
        f.write("%s: HBM-side bytes per call %.1f MB = %.2f x the algorithmic %.1f MB\n" % (impl, b / 1e6, b / alg, alg / 1e6))
PY
cat $out
