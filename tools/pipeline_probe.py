#!/usr/bin/env python3
"""Does overlapping the tail of one query's scan with the head of the next pay?  Two Searchers (two contexts, two
streams) take the 20 queries alternately vs one Searcher that runs them one after the other (peak DB)."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cudasw4_amd import capi, driver, search
_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
queries = [driver.encode(q) for q in letters]
sum_q = sum(len(q) for q in queries)
for num in (1_000_000, 125_000):
    L = 512
    db = search.DeviceDB.pseudo(num, L, driver.pseudo_sequence(L, 42), device=0)
    kt = search.KernelTypeConfig()
    ss = [search.Searcher(device=0, num_top=10, matrix=driver.matrix(62), kernel_types=kt) for _ in range(2)]
    prios = [int(x) for x in os.environ.get("PROBE_PRIOS", "0,-1").split(",")]
    streams = [torch.cuda.Stream(device=0, priority=prios[0]), torch.cuda.Stream(device=0, priority=prios[1])]
    for s in ss:
        s.set_database(db)
        s.scan(queries[0]); s.scan(queries[19])
    def run(mode):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "serial":
            for q in queries:
                ss[0].scan(q, timed=False)
        elif mode == "one-stream-async":
            pend = [ss[0].scan(q, timed=False, sync=False) for q in queries]
            torch.cuda.synchronize()
        else:
            pend = []
            for i, q in enumerate(queries):
                k = i % 2
                if len(pend) >= 2:
                    streams[k].synchronize()   # the slot's previous query is done
                with torch.cuda.stream(streams[k]):
                    pend.append(ss[k].scan(q, timed=False, sync=False))
            torch.cuda.synchronize()
        return time.perf_counter() - t0
    for mode in ("serial", "one-stream-async", "two-contexts", "serial", "two-contexts"):
        dt = run(mode)
        print("n=%d %-18s %.4f s  %.1f GCUPS" % (num, mode, dt, sum_q * num * L / 1e9 / dt))
