#!/usr/bin/env python3
"""Per-query, per-launch timing of the Swiss-Prot-like workload through both host drivers (diagnostics)."""
import os, sys, time, json
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cudasw4_amd import capi, driver, search, synthdb

chars, offsets, lengths = synthdb.sprot_like()
_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
queries = [driver.encode(q) for q in letters]
res = float(lengths.astype(np.int64).sum())
kinds = (1, 1, 2, 2)
top = int(os.environ.get("TOP", "10"))

d = driver.Driver(devices=[0], num_top=top, kinds=kinds)
d.db_from_arrays(chars, offsets, lengths)
d.upload()
for q in letters[:3]:
    d.scan(q)
d.record_kernel_events(True)
rows = []
t0 = time.perf_counter()
for q in letters:
    t = time.perf_counter()
    r = d.scan(q)
    rows.append((len(q), (time.perf_counter() - t) * 1e3, r["num_overflows"]))
dt = time.perf_counter() - t0
ev = d.take_kernel_events()
print("C++ driver: %.1f GCUPS total, %.3f s" % (sum(len(q) for q in letters) * res / 1e9 / dt, dt))
for (ql, ms, ovf) in rows:
    mine = [(e["part_id"], e["subjects"], round(e["ms"], 2)) for e in ev if e["qlen"] == ql]
    print("  q%5d  %8.2f ms  %7.1f GCUPS  ovf %d  launches %s" % (ql, ms, ql * res / 1e6 / ms, ovf, mine))
d.close()

db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
s = search.Searcher(device=0, num_top=top, matrix=driver.matrix(62), kernel_types=search.KernelTypeConfig.dpx())
s.set_database(db)
for q in queries[:3]:
    s.scan(q)
t0 = time.perf_counter()
rows = []
for q in queries:
    s.record_kernel_events = True
    s.kernel_events = []
    r = s.scan(q)
    rows.append((len(q), r.seconds * 1e3, [(run["part_id"], run["end"] - run["begin"], round(a.elapsed_time(b), 2)) for run, (a, b, _) in zip([s._plan[i] for i in ([i for i in range(len(s._plan)) if i != max(range(len(s._plan)), key=lambda i: s._plan[i]["end"] - s._plan[i]["begin"])] + [max(range(len(s._plan)), key=lambda i: s._plan[i]["end"] - s._plan[i]["begin"])])], s.kernel_events)]))
dt = time.perf_counter() - t0
print("Python mirror: %.1f GCUPS total, %.3f s" % (sum(len(q) for q in queries) * res / 1e9 / dt, dt))
for (ql, ms, runs) in rows:
    print("  q%5d  %8.2f ms  %7.1f GCUPS  launches %s" % (ql, ms, ql * res / 1e6 / ms, runs))
