#!/bin/bash
# Round-end evidence on the GPU box: bench line, rocprofv3 kernel stats of the SAME command, PMC HBM traffic.
# usage: scripts/round_profile.sh r01      -> gpurun_out/r01/{bench.json,kernel_stats.csv,FETCH_SIZE.summary.csv,...}
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out /tmp/$tag.ks
host_state() { echo "loadavg $(cat /proc/loadavg) | $(grep -E 'nr_throttled|throttled_usec' /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')"; }
host_state > $out/host_state.txt
python bench.py --no-cpu-baseline --no-full-graph --no-arm --steps 10 > /dev/null 2>&1   # page the image in; the first process on a fresh box is slow
host_state >> $out/host_state.txt
python bench.py > $out/bench.json 2> $out/bench.stderr
host_state >> $out/host_state.txt
tail -c 600 $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag.ks -o r -- python bench.py --no-cpu-baseline --no-full-graph --no-arm > $out/bench_under_rocprof.json 2> $out/rocprof.stderr
cp /tmp/$tag.ks/r_kernel_stats.csv $out/kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/$tag.$ctr && mkdir -p /tmp/$tag.$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/$tag.$ctr -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-graph --no-arm > /dev/null 2> $out/pmc.$ctr.stderr
  python scripts/pmc_summary.py /tmp/$tag.$ctr/r_counter_collection.csv $ctr > $out/$ctr.summary.csv
done
head -4 $out/FETCH_SIZE.summary.csv $out/WRITE_SIZE.summary.csv
python scripts/pmc_to_json.py $out $out/pmc_latest.json
