#!/usr/bin/env python3
"""The reference's runpeakbenchmark.sh as one script: every kernel type x pseudo-DB length, all 20
queries of allqueries.fasta, --top 0, DB resident (runpeakbenchmark.sh:26-83).  Prints per-(kernel, L)
total GCUPS (main.cu:257-260) and a per-query table for one chosen length.

    python tools/peak_sweep.py [--db-size 1000000] [--kernels half2,dpxs16,dpxs32,float] [--lengths 128,...]
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

KINDS = {"half2": 0, "dpxs16": 1, "dpxs32": 2, "float": 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--db-size", type=int, default=1_000_000)
    ap.add_argument("--kernels", default="half2,dpxs16,dpxs32,float")
    ap.add_argument("--lengths", default="128,256,512,768,1024,2048")
    ap.add_argument("--per-query-length", type=int, default=512)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    _, _letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
    queries = [driver.encode(q) for q in _letters]
    sum_q = sum(len(q) for q in queries)
    results = []
    for L in [int(x) for x in args.lengths.split(",")]:
        db = search.DeviceDB.pseudo(args.db_size, L, driver.pseudo_sequence(L, 42), device=0)
        for kname in args.kernels.split(","):
            kind = KINDS[kname]
            if kind in (2, 3) and L > 1024:
                continue  # runpeakbenchmark.sh:45,74: 32-bit kinds up to 1024
            big = capi.KIND_F32 if kind in (0, 3) else capi.KIND_I32
            small = kind if kind in (0, 1) else (0 if kind == 3 else 1)
            kt = search.KernelTypeConfig(kind, small, big, big)
            s = search.Searcher(device=0, num_top=0, matrix=driver.matrix(62), kernel_types=kt)
            s.set_database(db)
            s.scan(queries[0])  # warm-up
            per_q = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for q in queries:
                if L == args.per_query_length:
                    r = s.scan(q)
                    per_q.append(round(r.gcups, 1))
                else:
                    s.scan(q, timed=False, sync=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            gcups = sum_q * args.db_size * L / 1e9 / dt
            rec = {"kernel": kname, "L": L, "gcups": round(gcups, 1), "seconds": round(dt, 4)}
            if per_q:
                rec["per_query_gcups"] = per_q
            results.append(rec)
            print(json.dumps(rec), flush=True)
            del s
        del db
        torch.cuda.empty_cache()
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"db_size": args.db_size, "query_lengths": [len(q) for q in queries], "results": results}, f, indent=1)


if __name__ == "__main__":
    main()
