#!/usr/bin/env python3
"""Per-query time of the Swiss-Prot-like DB resident vs streamed through the C++ driver, with the streamed scan's batch
intervals (begin/end ms relative to the scan start): where does a streamed scan lose time?  Every configuration runs in
a process of its own (which streams share a hardware queue depends on what the process created before).

  python tools/stream_diag.py [config index]"""
import os, sys, time, subprocess
CONFIGS = (("resident", {}), ("streamed 400M", dict(max_gpu_mem=400 << 20)),
           ("streamed 400M / 32M batches", dict(max_gpu_mem=400 << 20, max_batch_bytes=32 << 20)), ("streamed 1G", dict(max_gpu_mem=1 << 30)),
           ("resident", {}))
if len(sys.argv) < 2:
    for i in range(len(CONFIGS)):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(i)])
    sys.exit(0)
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from cudasw4_amd import driver, synthdb
_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
db = synthdb.sprot_like(synthdb.SPROT_SEQUENCES)
res = {}
for name, kw in (CONFIGS[int(sys.argv[1])],):
    d = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2), **kw)
    d.db_from_arrays(*db)
    d.upload()
    for q in letters[:3]:
        d.scan(q)
    t = []
    for q in letters:
        t0 = time.perf_counter()
        d.scan(q)
        t.append((time.perf_counter() - t0) * 1e3)
    res[name] = t
    print("%-28s total %.1f ms  per query: %s" % (name, sum(t), " ".join("%.1f" % x for x in t)))
    if kw:
        for qi in (0, 19):
            d.scan(letters[qi])
            iv = d.batch_intervals()
            print("   query %d batches (begin..end ms): %s" % (qi, " ".join("%.1f..%.1f" % (b, e) for _, b, e in iv)))
    d.close()
