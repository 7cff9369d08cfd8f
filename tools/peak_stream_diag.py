#!/usr/bin/env python3
"""Peak DB (10^6 x 512) streamed in 128 MB batches through the C++ driver: per-query ms and the batch intervals."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from cudasw4_amd import driver
_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
d = driver.Driver(devices=[0], num_top=10, kinds=(0, 0, 3, 3), max_gpu_mem=1 << 30)
d.pseudo_db(1_000_000, 512)
d.upload()
for q in letters[:2] + letters[8:9]:
    d.scan(q)
t = []
for q in letters:
    t0 = time.perf_counter(); d.scan(q); t.append((time.perf_counter() - t0) * 1e3)
print("%s total %.1f ms  per query: %s" % (os.environ.get("CUDASW4_AMD_ONE_WORK_STREAM", "0"), sum(t), " ".join("%.1f" % x for x in t)))
for qi in (0, 5, 19):
    d.scan(letters[qi])
    print("   query %d batches: %s" % (qi, " ".join("%.1f..%.1f" % (b, e) for _, b, e in d.batch_intervals())))
