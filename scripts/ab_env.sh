#!/bin/bash
# Same-box alternation of one environment switch: scripts/ab_env.sh VAR=a VAR=b [rounds] [bench flags...]
# prints ms/step and host issue time of `bench.py --no-cpu-baseline --no-full-graph --no-arm --steps 30` per leg.
a=$1; b=$2; rounds=${3:-3}; shift 3
python bench.py --no-cpu-baseline --no-full-graph --no-arm --steps 10 > /dev/null 2>&1   # page the image in
for r in $(seq $rounds); do
  for leg in "$a" "$b"; do
    env $leg python bench.py --no-cpu-baseline --no-full-graph --no-arm --steps 30 "$@" 2> /dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$leg', round(d['ms_per_step'],3), 'host', d['host_issue_ms_per_step'])"
  done
done
cat /proc/loadavg
