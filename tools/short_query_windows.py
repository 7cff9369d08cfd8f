#!/usr/bin/env python3
"""Short queries on the Swiss-Prot-like DB with and without the exact windowing of the long subjects
(include/cudasw4_amd.h: sw_window_overlap; CUDASW4_AMD_NO_WINDOWS=1 turns it off): whole-scan GCUPS per query length,
through the C++ host driver, best of 3.
    python tools/short_query_windows.py [--lengths 48,96,144,189,222,300,375] [--configs dpx,half2]"""
import argparse, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from cudasw4_amd import driver, synthdb

ap = argparse.ArgumentParser()
ap.add_argument("--lengths", default="48,96,144,189,222,300,375,464")
ap.add_argument("--configs", default="dpx,half2")
args = ap.parse_args()
chars, offsets, lengths = synthdb.sprot_like()
residues = float(lengths.astype(np.int64).sum())
rng = np.random.default_rng(1)
alphabet = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
queries = [alphabet[rng.integers(0, 20, n)].tobytes() for n in (int(x) for x in args.lengths.split(","))]
CONFIGS = {"dpx": (1, 1, 2, 2), "half2": (0, 0, 3, 3)}
for cname in args.configs.split(","):
    rows = {}
    for mode in ("1", "0"):
        os.environ["CUDASW4_AMD_NO_WINDOWS"] = mode
        d = driver.Driver(devices=[0], num_top=10, kinds=CONFIGS[cname])
        d.db_from_arrays(chars, offsets, lengths)
        d.upload()
        d.scan(queries[0])
        out, tops = [], []
        for q in queries:
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                r = d.scan(q)
                best = min(best, time.perf_counter() - t0)
            out.append(len(q) * residues / 1e9 / best)
            tops.append((r["scores"].tolist(), r["ids"].tolist()))
        rows[mode] = (out, tops, d.window_stats())
        d.close()
    assert rows["0"][1] == rows["1"][1], "windows changed a result"
    print("%s: query residues        %s" % (cname, " ".join("%7d" % len(q) for q in queries)))
    print("%s: unsplit giants, GCUPS  %s" % (cname, " ".join("%7.0f" % v for v in rows["1"][0])))
    print("%s: windows, GCUPS         %s   (%d window launches, %d windows)" % (cname, " ".join("%7.0f" % v for v in rows["0"][0]), *rows["0"][2]))
