"""GPU idle time inside steady-state steps from a rocprofv3 --kernel-trace CSV: union of kernel intervals over ALL
queues vs wall time, per-queue busy time, and the gap histogram of the busiest queue.
    python scripts/trace_gaps.py <kernel_trace.csv> [last_ms]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
rows.sort()
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
t0, t1 = rows[0][0], rows[-1][1]
lo = t1 - last_ms * 1e6             # steady state: the last `last_ms` milliseconds of the trace
rows = [r for r in rows if r[0] >= lo]
wall = rows[-1][1] - rows[0][0]
# union over all queues
busy, cur_s, cur_e = 0, None, None
for s, e, q, n in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels %d  wall %.2f ms  GPU busy (union of all queues) %.2f ms = %.1f %%  idle %.2f ms" % (
    len(rows), wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6))
per_q = defaultdict(list)
for s, e, q, n in rows:
    per_q[q].append((s, e, n))
for q, lst in sorted(per_q.items(), key=lambda kv: -len(kv[1])):
    b = sum(e - s for s, e, _ in lst)
    print("  queue %s: %d kernels, busy %.2f ms (%.1f %% of wall), mean kernel %.1f us" % (q, len(lst), b / 1e6, 100.0 * b / wall, b / len(lst) / 1e3))
q, lst = max(per_q.items(), key=lambda kv: len(kv[1]))
gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
pos = [g for g in gaps if g > 0]
print("busiest queue %s: %d gaps > 0, total %.2f ms; median %.2f us, mean %.2f us" % (q, len(pos), sum(pos) / 1e6, sorted(pos)[len(pos) // 2] / 1e3, sum(pos) / len(pos) / 1e3))
for lim in (1, 2, 5, 10, 50, 200, 1e9):
    sel = [g for g in pos if g <= lim * 1e3]
    print("   gaps <= %6g us: %5d, %.2f ms" % (lim, len(sel), sum(sel) / 1e6))
big = sorted(((lst[i + 1][0] - lst[i][1], lst[i][2][:60], lst[i + 1][2][:60]) for i in range(len(lst) - 1)), reverse=True)[:12]
for g, a, b in big:
    print("   %.1f us between %s -> %s" % (g / 1e3, a, b))
