#!/usr/bin/env python3
"""Timeline of the DP launches of a rocprofv3 --kernel-trace run (rocpd sqlite): start / end / queue per launch, the overlap
of consecutive launches and the idle gaps between them.   python tools/kernel_timeline.py results.db [max rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
q = "select name, start, end, queue_id, stream_id, grid_x from kernels where name like '%sw_scan_kernel%' order by start"
try:
    rows = list(db.execute(q))
except sqlite3.OperationalError:
    rows = [r + (0,) for r in db.execute(q.replace("stream_id, ", ""))]
    rows = [(r[0], r[1], r[2], r[3], 0, r[4]) for r in rows]
t0 = rows[0][1]
prev_end = None
busy_end = 0
for name, s, e, qid, sid, grid in rows[:limit]:
    short = name[name.index("<"):name.index(">") + 1]
    gap = (s - busy_end) / 1e3 if busy_end else 0.0
    print("%-28s q%-3s s%-3s grid %5d  start %10.1f us  dur %9.1f us  start-vs-busy-end %8.1f us" % (short, qid, sid, grid, (s - t0) / 1e3, (e - s) / 1e3, gap))
    busy_end = max(busy_end, e)

# summary over the second half of the launches (the timed step of `bench.py --steps 1 --warmup 1`)
half = rows[len(rows) // 2:]
iv = sorted((s, e) for _, s, e, _, _, _ in half)
span = (max(e for _, e in iv) - iv[0][0]) / 1e6
union, cb, ce = 0, None, None
for s, e in iv:
    if ce is None or s > ce:
        if ce is not None:
            union += ce - cb
        cb, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cb
print("second half: %d launches, span %.3f ms, union busy %.3f ms, sum of durations %.3f ms" % (
    len(half), span, union / 1e6, sum(e - s for s, e in iv) / 1e6))
