#!/usr/bin/env python3
"""Timeline of one pass of allqueries.fasta over a 1/N shard of the Swiss-Prot-like DB through the C++ driver, walked the way
`align` walks a query file (Driver.scan_stream: two queries in flight where the driver's rule says so): every DP launch with
its begin and end on the device clock (HIP events), per query the span of its launches and the cells it scanned, and where
the GPU waited between queries.   python tools/shard_timeline.py [denominator=8] [--half2]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cudasw4_amd import driver, synthdb
denom = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
kinds = (0, 0, 3, 3) if "--half2" in sys.argv else (1, 1, 2, 2)
chars, offsets, lengths = synthdb.sprot_like(synthdb.SPROT_SEQUENCES // denom)
residues = float(lengths.astype(np.int64).sum())
_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
d = driver.Driver(devices=[0], num_top=10, kinds=kinds)
d.db_from_arrays(chars, offsets, lengths)
d.upload()
for q in letters:
    d.scan(q)
d.scan_stream(letters)
best, ev_best = 1e9, None
for _ in range(3):
    d.record_kernel_events(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = d.scan_stream(letters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d.record_kernel_events(False)
    ev = d.take_kernel_events()
    if dt < best:
        best, ev_best, res_best = dt, ev, res
cells = sum(len(q) for q in letters) * residues
print("1/%d shard: %d subjects, %.0f residues; pass %.2f ms, %.0f GCUPS; two in flight: %s" % (
    denom, len(lengths), residues, best * 1e3, cells / 1e9 / best, d.prefers_two_in_flight()))
ev = sorted(ev_best, key=lambda e: e["t0_ms"])
base = ev[0]["t0_ms"]
qlens = [len(q) for q in letters]
busy_end = 0.0
for e in ev:
    qi = qlens.index(int(e["qlen"])) if int(e["qlen"]) in qlens else -1
    gap = e["t0_ms"] - base - busy_end
    print("q%-2d len %4d  p%-2d %s R%dx%d%s n=%-6d %s [%8.2f .. %8.2f] %7.2f ms%s" % (
        qi, e["qlen"], e["part_id"], ["f16", "i16", "i32", "f32"][e["eff_kind"]], e["rows"], e["lanes"], "m" if e["nstripes"] > 1 else " ",
        e["subjects"], "rescore" if e["rescore"] else "scan   ", e["t0_ms"] - base, e["t1_ms"] - base, e["ms"],
        "   <- nothing ran for %.2f ms" % gap if gap > 0.02 else ""))
    busy_end = max(busy_end, e["t1_ms"] - base)
print("launch span %.2f ms of the pass's %.2f ms" % (busy_end, best * 1e3))
for qi, q in enumerate(letters):
    mine = [e for e in ev if int(e["qlen"]) == len(q)]
    if not mine:
        continue
    b, e = min(x["t0_ms"] for x in mine) - base, max(x["t1_ms"] for x in mine) - base
    bulk = max(mine, key=lambda x: x["subjects"])
    print("q%-2d len %4d: launches %8.2f .. %8.2f (%6.2f ms), bulk launch %6.2f ms = %5.0f GCUPS in itself; ideal at 11.4 TCUPS %5.2f ms; %d overflows, %d re-scored" % (
        qi, len(q), b, e, e - b, bulk["ms"], len(q) * residues / 1e6 / bulk["ms"], len(q) * residues / 11.4e9,
        res_best[qi]["num_overflows"], res_best[qi]["num_rescored"]))
