#!/usr/bin/env python3
"""Speed of the DOCUMENTED boundary against the host driver on the same DBs.

tests/boundary/_build/binding_gpu is the reference-side binding — the verbatim code block of INTEGRATION.md, compiled against
the reference's own headers — with a --bench mode: every query of allqueries.fasta through the documented per-query block
(sw_set_query + sw_scan_batch + sw_batch_join + sw_topk), and through the launcher-by-launcher form (one sw_scan_partition per
length partition, sw_rescore_overflow, sw_topk) beside it.  The host driver (libcudasw4_host.so, what bench.py measures) scans
the same arrays in this process.  GCUPS over the 20 queries, best of `reps` passes, top-K inside.

    python tools/binding_bench.py [--peak-size 1000000] [--sprot-size 570000] [--reps 2]
"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "tests", "boundary", "_build", "binding_gpu")


def write_db(prefix, chars, offsets, lengths):
    np.ascontiguousarray(chars, dtype=np.int8).tofile(prefix + "0chars")
    np.ascontiguousarray(offsets, dtype=np.uint64).tofile(prefix + "0offsets")
    np.ascontiguousarray(lengths, dtype=np.int32).tofile(prefix + "0lengths")


def write_queries(path, encoded):
    with open(path, "wb") as f:
        for q in encoded:
            f.write(struct.pack("<i", len(q)))
            f.write(np.ascontiguousarray(q, dtype=np.int8).tobytes())


def run_binding(prefix, qfile, dpx, reps):
    p = subprocess.run([EXE, prefix, "0", "DPX" if dpx else "-", "--bench", qfile, str(reps)], capture_output=True, text=True, timeout=1800)
    if p.returncode != 0 or "binding ok" not in p.stdout:
        raise RuntimeError("binding_gpu failed: " + p.stderr[-2000:] + p.stdout[-500:])
    out = {}
    for line in p.stdout.splitlines():
        if line.startswith("BENCH "):
            parts = line.split()
            kv = dict(x.split("=") for x in parts[2:5])
            out[parts[1]] = {"gcups": float(kv["gcups"]), "seconds": float(kv["seconds"]), "top1": [int(x) for x in parts[parts.index("TOP1") + 1:]]}
    return out


def run_driver(arrays, letters, kinds, reps, pseudo=None):
    import torch
    from cudasw4_amd import driver
    d = driver.Driver(devices=[0], num_top=5, kinds=kinds)
    if pseudo:
        d.pseudo_db(*pseudo)
        residues = float(pseudo[0]) * pseudo[1]
    else:
        d.db_from_arrays(*arrays)
        residues = float(arrays[2].astype(np.int64).sum())
    d.upload()
    res = d.scan_stream(letters)
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = d.scan_stream(letters)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    d.close()
    return {"gcups": sum(len(q) for q in letters) * residues / 1e9 / best, "seconds": best, "top1": [int(r["scores"][0]) for r in res]}


def bench(peak_size, sprot_size, reps, out=sys.stdout):
    from cudasw4_amd import driver, synthdb
    _, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
    encoded = [driver.encode(q) for q in letters]
    rows = []
    with tempfile.TemporaryDirectory() as td:
        qfile = os.path.join(td, "queries.bin")
        write_queries(qfile, encoded)
        # the peak DB: one pseudo sequence, peak_size times (dbdata layout)
        codes = driver.pseudo_sequence(512, 42)
        chars = np.tile(codes, peak_size)
        offsets = np.arange(peak_size + 1, dtype=np.uint64) * 512
        lengths = np.full(peak_size, 512, dtype=np.int32)
        legs = [("peak %d x 512, half2" % peak_size, (chars, offsets, lengths), False, (0, 0, 3, 3))]
        sp = synthdb.sprot_like(sprot_size)
        legs.append(("Swiss-Prot-like %d subjects, dpx" % sprot_size, sp, True, (1, 1, 2, 2)))
        for name, arrays, dpx, kinds in legs:
            prefix = os.path.join(td, "db")
            write_db(prefix, *arrays)
            b = run_binding(prefix, qfile, dpx, reps)
            d = run_driver(arrays, letters, kinds, reps)
            ok = b["sw_scan_batch"]["top1"] == d["top1"] == b["one_launch_per_partition"]["top1"]
            rows.append((name, b["sw_scan_batch"]["gcups"], b["one_launch_per_partition"]["gcups"], d["gcups"], ok))
            out.write("%-44s binding (sw_scan_batch) %8.1f GCUPS   one launch per partition %8.1f   host driver %8.1f   ratio %.3f   top scores equal: %s\n" % (
                name, b["sw_scan_batch"]["gcups"], b["one_launch_per_partition"]["gcups"], d["gcups"], b["sw_scan_batch"]["gcups"] / d["gcups"], ok))
            out.flush()
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--peak-size", type=int, default=1_000_000)
    ap.add_argument("--sprot-size", type=int, default=570_000)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    bench(a.peak_size, a.sprot_size, a.reps)
