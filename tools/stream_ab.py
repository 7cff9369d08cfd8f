#!/usr/bin/env python3
"""Per-query A/B of the streamed-subjects kernel (CUDASW4_AMD_STREAM) on the Swiss-Prot-like DB, single-stripe queries."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from cudasw4_amd import driver, synthdb

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 570000
    kinds = (1, 1, 2, 2)
    chars, offsets, lengths = synthdb.sprot_like(n)
    _, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
    res = {}
    for st in sys.argv[2:] or ["0", "4"]:
        os.environ["CUDASW4_AMD_STREAM"] = st
        d = driver.Driver(devices=[0], num_top=10, kinds=kinds)
        d.db_from_arrays(chars, offsets, lengths)
        d.upload()
        for qi in range(10):
            q = letters[qi]
            d.scan(q)
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter(); d.scan(q); best = min(best, time.perf_counter() - t0)
            res[(st, qi)] = best
        d.close()
    tot = float(lengths.astype(np.int64).sum())
    for qi in range(10):
        print("query %4d: " % len(letters[qi]) + "  ".join("STREAM=%s %.3f ms %.0f GCUPS" % (st, res[(st, qi)] * 1e3, len(letters[qi]) * tot / res[(st, qi)] / 1e9)
                                                          for st in (sys.argv[2:] or ["0", "4"])))
main()
