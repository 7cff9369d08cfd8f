#!/usr/bin/env python3
"""Timeline of ALL DP launches (scan, row-parallel, pipelined) of a rocprofv3 --kernel-trace run (rocpd sqlite), for the
last `nq` queries:   python tools/timeline_all.py results.db [rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 80
q = ("select name, start, end, queue_id, grid_x from kernels where name like '%sw_scan_kernel%' or name like '%sw_rows%' "
     "or name like '%fill%' or name like '%topk%' order by start")
rows = list(db.execute(q))
rows = rows[-limit:]
t0 = rows[0][1]
for name, s, e, qid, grid in rows:
    short = name[:name.index("(")] if "(" in name else name
    short = short.replace("swk::", "").replace("void ", "")[:70]
    print("%-70s q%-3s grid %6d  start %10.1f us  end %10.1f us  dur %9.1f us" % (short, qid, grid, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
