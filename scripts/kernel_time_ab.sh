#!/bin/bash
# A/B a switch by total kernel time per step under rocprofv3 (wall time varies +-2 % run to run on a box).
# usage: scripts/kernel_time_ab.sh ENV_VAR [steps]     runs bench.py with ENV_VAR=0 and ENV_VAR=1
var=$1; steps=${2:-10}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  rm -rf /tmp/ab.$v && mkdir -p /tmp/ab.$v
  env $var=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab.$v -o r -- python bench.py --no-cpu-baseline --steps $steps --warmup 3 > /tmp/ab.$v/bench.json 2> /tmp/ab.$v/err
  python - $v $steps /tmp/ab.$v/r_kernel_stats.csv "$var" <<'PY'
import csv, sys
v, steps, path, var = sys.argv[1], int(sys.argv[2]) + 3, sys.argv[3], sys.argv[4]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
n = sum(int(r["Calls"]) for r in rows) / steps
print("%s=%s: %.3f ms of kernels / step, %.0f launches / step" % (var, v, tot, n))
for r in rows:
    if any(k in r["Name"] for k in ("colsum", "reduce_kernel")):
        print("    %-60s %6.1f calls/step %8.3f ms/step" % (r["Name"][:60], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps))
PY
done
