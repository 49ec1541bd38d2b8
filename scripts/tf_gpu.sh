set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_trajectoryformer_golden.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/tf_test.log
timeout 600 python bench.py --model trajectoryformer --scenes 4 --steps 10 --warmup 3 > gpurun_out/tf_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tf_prof -o tf -- python $GRAFT_REPO_ROOT/bench.py --model trajectoryformer --scenes 4 --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/tf_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/tf_prof -type f ! -name '*stats.csv' -delete
python - <<'PY' > gpurun_out/tf_prof_top.txt 2>&1
import csv, glob
f = glob.glob('gpurun_out/tf_prof/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print('total GPU ms', tot/1e6, 'calls', calls)
for r in rows[:40]:
    print('%-90s %6s %10.3f ms %6.2f%%' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['Percentage'])))
PY
cat gpurun_out/tf_test.log gpurun_out/tf_bench.log | tail -30
