#!/usr/bin/env python3
"""The giants' kernels alone: one (or a few) very long subject(s) against queries of several lengths, timed with events.

    python tools/giants_bench.py [--subject 35213] [--nsubjects 1] [--reps 5]

Prints ms per launch for sw_scan_rows_pipelined at 4 / 8 / 16 columns per lane
(CUDASW4_AMD_PIPE_CPL), and checks that all of them return the same scores."""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from cudasw4_amd import capi, search, driver


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--subject", type=int, default=35213)
    ap.add_argument("--nsubjects", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--queries", default="48,144,567,1000,2005,5478")
    args = ap.parse_args()
    rng = np.random.default_rng(1)
    lens = np.sort(np.linspace(args.subject, max(8001, args.subject // 3), args.nsubjects).astype(np.int64))
    seqs = [rng.integers(0, 20, int(l)).astype(np.int8) for l in lens]
    padded = [(len(s) + 3) // 4 * 4 for s in seqs]
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(padded)
    chars = np.full(int(offsets[-1]), 20, dtype=np.int8)
    for s, o in zip(seqs, offsets[:-1]):
        chars[int(o):int(o) + len(s)] = s
    lengths = np.array([len(s) for s in seqs], dtype=np.int32)
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    n = len(seqs)
    maxlen = int(lengths.max())
    m = driver.matrix(62)
    ctxs = {}
    for cpl in (4, 8, 16):
        if cpl:
            os.environ["CUDASW4_AMD_PIPE_CPL"] = str(cpl)
        ctxs[cpl] = capi.Context(0)
        ctxs[cpl].set_matrix(m)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    print("subjects: %d, longest %d residues" % (n, maxlen))
    for qlen in [int(x) for x in args.queries.split(",")]:
        q = rng.integers(0, 20, qlen).astype(np.int8)
        row = []
        ref = None
        for cpl in (4, 8, 16):
            ctx = ctxs[cpl]
            ctx.set_query(q)
            scores = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
            ids = torch.full((n,), -1, dtype=torch.int32, device="cuda")
            if True:
                tb = ctx.scan_rows_pipelined_temp_bytes(n, maxlen)
                temp = torch.empty(tb, dtype=torch.uint8, device="cuda")
                fails = torch.zeros(1, dtype=torch.int32, device="cuda")
                run = lambda: ctx.scan_rows_pipelined(db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(), 0, n, maxlen,
                                                      -11, -1, scores.data_ptr(), ids.data_ptr(), 0, fails.data_ptr(), temp.data_ptr(), tb)
            run()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(args.reps):
                ev0.record()
                run()
                ev1.record()
                torch.cuda.synchronize()
                best = min(best, ev0.elapsed_time(ev1))
            got = scores.cpu().numpy()
            if ref is None:
                ref = got
            assert (got == ref).all(), (qlen, cpl, got, ref)
            cells = float(qlen) * float(lengths.astype(np.int64).sum())
            row.append("%s: %.3f ms (%.1f GCUPS, %.3f us/row)" % ("rows" if cpl == 0 else "pipe%d" % cpl, best, cells / best / 1e6, best * 1e3 / qlen))
        print("query %5d | " % qlen + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
