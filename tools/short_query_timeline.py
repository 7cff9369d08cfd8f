#!/usr/bin/env python3
"""One short query at a time on the Swiss-Prot-like DB through the C++ driver — run under `rocprofv3 --kernel-trace` to see
what a query's fixed work consists of (tools/timeline_all.py prints the launches).   python tools/short_query_timeline.py [qlen] [reps]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from cudasw4_amd import driver, synthdb
qlen = int(sys.argv[1]) if len(sys.argv) > 1 else 48
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
chars, offsets, lengths = synthdb.sprot_like()
rng = np.random.default_rng(1)
alphabet = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
qs = [alphabet[rng.integers(0, 20, qlen)].tobytes() for _ in range(reps)]
d = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2))
d.db_from_arrays(chars, offsets, lengths)
d.upload()
for q in qs[:2]:
    d.scan(q)
t0 = time.perf_counter()
for q in qs:
    d.scan(q)
dt = (time.perf_counter() - t0) / reps
print("query %d residues alone: %.3f ms per scan, %.0f GCUPS" % (qlen, dt * 1e3, qlen * float(lengths.astype(np.int64).sum()) / dt / 1e9))
t0 = time.perf_counter()
d.scan_many(qs)
dt = (time.perf_counter() - t0) / reps
print("stream of %d: %.3f ms per scan, %.0f GCUPS" % (reps, dt * 1e3, qlen * float(lengths.astype(np.int64).sum()) / dt / 1e9))
d.close()
