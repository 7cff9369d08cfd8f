#!/usr/bin/env python3
"""Where do the side launches of a resident Swiss-Prot-like scan run?  For several stream creation orders
(CUDASW4_AMD_STREAM_ORDER) the launches of the longest queries with their begin / end on the device clock (HIP events on
the stream each launch ran on): beside the bulk launch, in front of it, or behind it.
    python tools/stream_order_probe.py [orders ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from cudasw4_amd import driver, synthdb
    orders = sys.argv[1:] or ["WCAB", "ABWC", "BAWC", "WACB"]
    _, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
    db = synthdb.sprot_like()
    residues = float(db[2].astype(np.int64).sum())
    for order in orders:
        os.environ["CUDASW4_AMD_STREAM_ORDER"] = order
        d = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2))
        d.db_from_arrays(*db)
        d.upload()
        for q in letters[-3:]:
            d.scan(q)
        t0 = time.perf_counter()
        for q in letters:
            d.scan(q)
        dt = time.perf_counter() - t0
        print("order %s: %.0f GCUPS over the 20 queries" % (order, sum(len(q) for q in letters) * residues / 1e9 / dt))
        for qi in (19, 10, 3):
            d.record_kernel_events(True)
            t0 = time.perf_counter()
            d.scan(letters[qi])
            wall = (time.perf_counter() - t0) * 1e3
            d.record_kernel_events(False)
            ev = d.take_kernel_events()
            print("  query %d (%d residues): scan %.2f ms" % (qi, len(letters[qi]), wall))
            for e in sorted(ev, key=lambda e: e["t0_ms"]):
                print("    %-8s part %2d lanes %2d R %2d x %d stripes  %7d subjects  [%8.3f, %8.3f] ms  (%.3f)" % (
                    "rescore" if e["rescore"] else "scan", e["part_id"], e["lanes"], e["rows"], e["nstripes"], e["subjects"],
                    e["t0_ms"], e["t1_ms"], e["ms"]))
        d.close()


if __name__ == "__main__":
    main()
