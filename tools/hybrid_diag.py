#!/usr/bin/env python3
"""Per-query time of the Swiss-Prot-like DB under a memory limit (hybrid residency: the longest subjects cached, the rest
streamed in batches) against resident, and every launch of two queries with begin / end (HIP events).
    python tools/hybrid_diag.py [max_gpu_mem=260M] [max_batch_bytes=16M]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from cudasw4_amd import driver, synthdb
mem = int(float(sys.argv[1]) * (1 << 20)) if len(sys.argv) > 1 else 260 << 20
bb = int(float(sys.argv[2]) * (1 << 20)) if len(sys.argv) > 2 else 16 << 20
_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
db = synthdb.sprot_like(synthdb.SPROT_SEQUENCES)
residues = float(db[2].astype(np.int64).sum())
for name, kw in (("resident", {}), ("hybrid", dict(max_gpu_mem=mem, max_batch_bytes=bb))):
    d = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2), **kw)
    d.db_from_arrays(*db)
    d.upload()
    for q in letters[:3]:
        d.scan(q)
    t = []
    for q in letters:
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            d.scan(q)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        t.append(best)
    print("%-10s %s total %.1f ms = %.0f GCUPS  per query: %s" % (name, d.shard_info(), sum(t), sum(len(q) for q in letters) * residues / 1e6 / sum(t), " ".join("%.1f" % x for x in t)))
    if kw:
        for qi in (5, 19):
            d.record_kernel_events(True)
            d.scan(letters[qi])
            d.record_kernel_events(False)
            ev = sorted(d.take_kernel_events(), key=lambda e: e["t0_ms"])
            base = ev[0]["t0_ms"]
            print("  query %d (%d residues):" % (qi, len(letters[qi])))
            for e in ev:
                print("    p%-2d %s R%dx%d%s n=%-6d %s [%7.2f .. %7.2f] %6.2f ms" % (e["part_id"], ["f16", "i16", "i32", "f32"][e["eff_kind"]], e["rows"], e["lanes"],
                      "m" if e["nstripes"] > 1 else " ", e["subjects"], "rescore" if e["rescore"] else "scan   ", e["t0_ms"] - base, e["t1_ms"] - base, e["ms"]))
    d.close()
