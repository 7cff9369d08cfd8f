#!/usr/bin/env python3
"""One query of allqueries.fasta against the pseudo DB, a few repetitions: the unit for kernel-level profiling.

    python tools/one_query.py --query-index 0 --length 128 --kernel half2 --reps 5
"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import torch

from cudasw4_amd import capi, driver, search

KINDS = {"half2": 0, "dpxs16": 1, "dpxs32": 2, "float": 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--query-index", type=int, default=0)
    ap.add_argument("--query-length", type=int, default=0, help="random query of this length instead")
    ap.add_argument("--length", type=int, default=128)
    ap.add_argument("--db-size", type=int, default=1_000_000)
    ap.add_argument("--kernel", default="half2")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    _, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
    q = driver.encode(letters[args.query_index])
    if args.query_length:
        q = driver.pseudo_sequence(args.query_length, 7)
    kind = KINDS[args.kernel]
    big = capi.KIND_F32 if kind in (0, 3) else capi.KIND_I32
    small = kind if kind in (0, 1) else (0 if kind == 3 else 1)
    kt = search.KernelTypeConfig(single_pass=kind, many_pass_small=small, many_pass_large=big, overflow=big)
    db = search.DeviceDB.pseudo(args.db_size, args.length, driver.pseudo_sequence(args.length, 42), device=0)
    s = search.Searcher(device=0, num_top=0, matrix=driver.matrix(62), kernel_types=kt)
    s.set_database(db)
    s.record_kernel_events = False
    s.scan(q, timed=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        s.scan(q, timed=False, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    print("query %d residues, %d x %d: %.3f ms, %.1f GCUPS" % (len(q), args.db_size, args.length, dt * 1e3,
                                                                 len(q) * args.db_size * args.length / dt / 1e9))


if __name__ == "__main__":
    main()
