#!/usr/bin/env python3
"""Per query of allqueries.fasta on the Swiss-Prot-like DB (C++ driver, resident, --dpx): whole-scan GCUPS and the duration
of every DP launch of the scan (bulk run, partition 34, partition 35 = the giants), from HIP events: which launch is the
critical path of which query."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from cudasw4_amd import driver, synthdb
SIZE = int(os.environ.get("BREAKDOWN_DB_SIZE", synthdb.SPROT_SEQUENCES))  # a smaller DB of the same length distribution = a shard
chars, offsets, lengths = synthdb.sprot_like(SIZE)
residues = float(lengths.astype(np.int64).sum())
_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
kinds = (1, 1, 2, 2) if "--half2" not in sys.argv else (0, 0, 3, 3)
d = driver.Driver(devices=[0], num_top=10, kinds=kinds)
d.db_from_arrays(chars, offsets, lengths)
d.upload()
d.scan(letters[0])
tot_t, tot_c = 0.0, 0.0
ONLY = [int(x) for x in os.environ.get("BREAKDOWN_QUERIES", "").split(",") if x]
for qi, q in enumerate(letters):
    if ONLY and qi not in ONLY:
        continue
    best, ev_best = 1e9, None
    for _ in range(3):
        d.record_kernel_events(True)
        s = d.scan(q)["seconds"]
        d.record_kernel_events(False)
        ev = d.take_kernel_events()
        if s < best:
            best, ev_best = s, ev
    tot_t += best
    tot_c += len(q) * residues
    span = max(e["t1_ms"] for e in ev_best) - min(e["t0_ms"] for e in ev_best)
    parts = ", ".join("p%d %s R%dx%d%s n=%d: %.2f ms [%.2f..%.2f]" % (e["part_id"], ["f16", "i16", "i32", "f32"][e["eff_kind"]], e["rows"], e["lanes"],
                      "m" if e["nstripes"] > 1 else "", e["subjects"], e["ms"], e["t0_ms"] - min(x["t0_ms"] for x in ev_best), e["t1_ms"] - min(x["t0_ms"] for x in ev_best))
                      for e in sorted(ev_best, key=lambda e: -e["ms"]))
    print("q%-2d len %4d  %6.0f GCUPS  scan %.2f ms  launches span %.2f ms | %s" % (qi, len(q), len(q) * residues / 1e9 / best, best * 1e3, span, parts))
print("all 20: %.0f GCUPS" % (tot_c / 1e9 / tot_t))
