#!/usr/bin/env python3
"""Swiss-Prot-like synthetic DB benchmark (stand-in for runsprotbenchmark.sh when the real
uniprot_sprot.fasta.gz is not available: no network on the build / GPU boxes).

Lengths: log-normal (median ~ 290, sigma 0.75) clipped to [2, 35213], N sequences (default 570000 ==
UniProtKB/Swiss-Prot), uniform random residues, seed 2024; sorted by length like makedb does.
Runs all 20 queries with the chosen kernel configuration and prints per-query and total GCUPS.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from cudasw4_amd import capi, driver, search


def synth_lengths(n, seed, max_len):
    rng = np.random.default_rng(seed)
    l = np.exp(rng.normal(np.log(290.0), 0.75, n)).astype(np.int64)
    l = np.clip(l, 2, max_len)
    # a realistic long tail: a few giant proteins (titin-like)
    k = max(1, n // 30000)
    l[:k] = np.linspace(max_len, 8000, k).astype(np.int64)
    return np.sort(l).astype(np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=570_000)
    ap.add_argument("--max-len", type=int, default=35213)
    ap.add_argument("--config", choices=["half2", "dpx"], default="dpx")
    ap.add_argument("--queries", default="all")
    ap.add_argument("--top", type=int, default=0)
    ap.add_argument("--runs", action="store_true", help="print per-launch times for three queries")
    args = ap.parse_args()
    lengths = synth_lengths(args.n, 2024, args.max_len)
    padded = (lengths.astype(np.int64) + 3) // 4 * 4
    offsets = np.zeros(len(lengths) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(padded)
    rng = np.random.default_rng(7)
    chars = rng.integers(0, 20, int(offsets[-1]), dtype=np.int8)
    # padding bytes -> 20
    pos = np.arange(int(offsets[-1]), dtype=np.int64)
    seq_of = np.searchsorted(offsets[1:].astype(np.int64), pos, side="right")
    chars[pos - offsets[seq_of].astype(np.int64) >= lengths[seq_of]] = 20
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    kt = search.KernelTypeConfig.dpx() if args.config == "dpx" else search.KernelTypeConfig()
    s = search.Searcher(device=0, num_top=args.top, matrix=driver.matrix(62), kernel_types=kt)
    s.set_database(db)
    _, _letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
    queries = [driver.encode(q) for q in _letters]
    if args.queries != "all":
        queries = [queries[int(i)] for i in args.queries.split(",")]
    print("DB: %d sequences, %d residues, max length %d; partitions used: %s" % (
        len(lengths), int(lengths.sum()), int(lengths.max()), [(r["part_id"], r["end"] - r["begin"]) for r in s._plan]))
    s.scan(queries[0])
    if args.runs:
        for qi in (0, 9, 19):
            s.record_kernel_events = True
            s.kernel_events = []
            r = s.scan(queries[qi])
            s.record_kernel_events = False
            print("query len %d total %.2f ms; runs (part, n, ms):" % (len(queries[qi]), r.seconds * 1e3),
                  [(run["part_id"], run["end"] - run["begin"], round(a.elapsed_time(b), 2)) for run, (a, b, _) in zip(s._plan, s.kernel_events)])
    per = []
    t0 = time.perf_counter()
    for q in queries:
        r = s.scan(q)
        per.append((len(q), round(r.seconds * 1e3, 2), round(r.gcups, 1)))
    dt = time.perf_counter() - t0
    total = sum(len(q) for q in queries) * float(lengths.sum()) / 1e9 / dt
    print(json.dumps({"config": args.config, "n": args.n, "total_gcups": round(total, 1), "seconds": round(dt, 3),
                      "per_query(len,ms,gcups)": per}))


if __name__ == "__main__":
    main()
