#!/usr/bin/env python3
"""Thread scaling of the CPU baseline (oracle's inter-sequence SIMD scan) on this host."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tests"))  # TEST INFRASTRUCTURE: times the oracle, nothing of the product
import oracle_lib as O

_, qs = O.load_queries()
codes = O.pseudodb_codes(512, 42)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
db = O.make_db([codes] * n)
q = qs[9]
print("host threads available:", O.max_threads(), "nproc:", os.cpu_count())
for nt in (1, 8, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1):
        break
    O.scan(q, *db, simd=True, nthreads=nt)
    t = time.perf_counter()
    O.scan(q, *db, simd=True, nthreads=nt)
    dt = time.perf_counter() - t
    print("%4d threads: %8.1f GCUPS (%.2f per thread)" % (nt, len(q) * n * 512 / 1e9 / dt, len(q) * n * 512 / 1e9 / dt / nt))
