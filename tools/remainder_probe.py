#!/usr/bin/env python3
"""How long do n equal-length subjects take on the standard group shape (8 / 16 lanes) and on wave-wide groups?
The question behind it (round 2): a launch of n subjects on W resident workgroups of S subjects each takes ceil(n / (W S))
rounds — does the last, under-filled round cost a whole round, and would a second launch on wave-wide groups (4x the lanes
per alignment, part_id 35) finish those n mod (W S) subjects sooner?

Answer (profiles/r02_remainder_probe.txt, MI355X): no.  A SIMD that holds ONE wave runs it 1.5x faster than each of two
(L = 512, 1000-residue query: 8 192 subjects = one workgroup per CU 0.58 ms, 16 384 = two per CU 0.87 ms), so an
under-filled round costs 2/3 of a full one up to half full, and inside a long launch the workgroups have drifted apart
by then, so the remainder overlaps the stragglers of the round before.  Wave-wide groups beat the standard shape only
below ~4 000 subjects per launch (0.3x at <= 2 048) and are 1.2 ... 2.5x slower per subject from 16 384 on.  A library
entry point that split every launch of nearly equal-length subjects into full rounds + a wave-wide remainder was built and
measured: 10^6 x 512 resident 11 499 -> 11 485 GCUPS, streamed 11 081 -> 11 052, a 125 000-subject shard 10 679 -> 10 563
— removed again (experiments).

  python tools/remainder_probe.py [--kind 0] [--lengths 128,512]"""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from cudasw4_amd import capi, driver, search

ap = argparse.ArgumentParser()
ap.add_argument("--kind", type=int, default=0)
ap.add_argument("--lengths", default="128,512")
ap.add_argument("--counts", default="64,576,2048,4096,8192,12288,16384,20000,32768")
ap.add_argument("--queries", default="0,2,5,8,9,11,15,19")
args = ap.parse_args()

_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
queries = [driver.encode(q) for q in letters]
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
ctx.set_matrix(driver.matrix(62))
stream = torch.cuda.current_stream().cuda_stream
counts = [int(x) for x in args.counts.split(",")]
nmax = max(counts)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for L in [int(x) for x in args.lengths.split(",")]:
    db = search.DeviceDB.pseudo(nmax, L, driver.pseudo_sequence(L, 42), device=0)
    scores = torch.zeros(nmax, dtype=torch.float32, device=dev)
    ids = torch.zeros(nmax, dtype=torch.int32, device=dev)
    ovf_pos = torch.zeros(nmax, dtype=torch.int32, device=dev)
    ovf_cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for qi in [int(x) for x in args.queries.split(",")]:
        q = queries[qi]
        ctx.set_query(q, stream)
        rows, ns = capi.plan_query(args.kind, len(q))
        print("L=%d query %d (%d residues; 16-lane plan: %d rows x %d stripes)" % (L, qi, len(q), rows, ns))
        for n in counts:
            res = {}
            for name, part in (("std", 0), ("wide", 35)):
                need = max(ctx.scan_temp_bytes(args.kind, part, n, L), 16)
                temp = torch.empty(need, dtype=torch.uint8, device=dev)

                def run():
                    ovf_cnt.zero_()
                    ctx.scan_partition(args.kind, part, db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(), 0, n, L,
                                       -11, -1, scores.data_ptr(), ids.data_ptr(), 0, ovf_pos.data_ptr(), ovf_cnt.data_ptr(), 1,
                                       temp.data_ptr(), need, stream)
                res[name] = timed(run)
            print("   n=%6d  std %8.3f ms   wide %8.3f ms   wide/std %.2f" % (n, res["std"], res["wide"], res["wide"] / res["std"]))
