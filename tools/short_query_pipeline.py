#!/usr/bin/env python3
"""Streams of SHORT queries on the Swiss-Prot-like DB through the C++ host driver: one query at a time (scan), two in
flight on one lane (scan_many, CUDASW4_AMD_TAIL_OVERLAP=0) and two in flight with the tail hand-over (the next query on the
GPU's second lane, CUDASW4_AMD_TAIL_OVERLAP=1).  A 48-residue query's scan takes 1.8 ms, 0.3 ms of it fixed work per query
(upload, profile builds, handshake, top-K, copy-back) that a second query in flight can hide.  GCUPS over the whole stream.
    python tools/short_query_pipeline.py [--lengths 48,96,144,189,222] [--copies 16] [--configs dpx,half2]"""
import argparse, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from cudasw4_amd import driver, synthdb

ap = argparse.ArgumentParser()
ap.add_argument("--lengths", default="48,96,144,189,222,300,375,464,567")
ap.add_argument("--copies", type=int, default=16)
ap.add_argument("--configs", default="dpx")
ap.add_argument("--db-size", type=int, default=synthdb.SPROT_SEQUENCES)
args = ap.parse_args()
chars, offsets, lengths = synthdb.sprot_like(args.db_size)
residues = float(lengths.astype(np.int64).sum())
rng = np.random.default_rng(1)
alphabet = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
CONFIGS = {"dpx": (1, 1, 2, 2), "half2": (0, 0, 3, 3)}
for cname in args.configs.split(","):
    table = {}
    for n in (int(x) for x in args.lengths.split(",")):
        queries = [alphabet[rng.integers(0, 20, n)].tobytes() for _ in range(args.copies)]
        row, tops = [], {}
        for mode, env in (("one at a time", None), ("two in flight, one lane", "0"), ("two in flight, two lanes", "1"),
                          ("two in flight, driver's rule", "auto")):
            os.environ.pop("CUDASW4_AMD_TAIL_OVERLAP", None)
            if env in ("0", "1"):
                os.environ["CUDASW4_AMD_TAIL_OVERLAP"] = env
            d = driver.Driver(devices=[0], num_top=10, kinds=CONFIGS[cname])
            d.db_from_arrays(chars, offsets, lengths)
            d.upload()
            d.scan(queries[0]); d.scan_many(queries[:3])
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                res = [d.scan(q) for q in queries] if env is None else d.scan_many(queries)
                best = min(best, time.perf_counter() - t0)
            tops[mode] = [(r["scores"].tolist(), r["ids"].tolist()) for r in res]
            row.append((mode, n * len(queries) * residues / 1e9 / best, d.tail_overlaps()))
            d.close()
        assert len({str(v) for v in tops.values()}) == 1, "modes disagree"
        table[n] = row
    print("%s: %d queries per length, GCUPS over the stream (same top-10 lists in every mode)" % (cname, args.copies))
    print("%-28s %s" % ("query residues", " ".join("%7d" % n for n in table)))
    for i in range(4):
        print("%-28s %s" % (table[next(iter(table))][i][0], " ".join("%7.0f" % table[n][i][1] for n in table)))
    print("%-28s %s" % ("  ... queries gated", " ".join("%7d" % table[n][3][2] for n in table)))
