#!/usr/bin/env python3
"""makedb at input scale (makedb.cpp:80-103, mmapbuffer.hpp:332-507): a FASTA of N sequences (default
2.5e7, log-normal lengths around 40 -> ~1.3e9 residues, 1.5 GB of text) under a --mem limit small enough that all five
arrays spill to their temp files, the parallel stable sort, and a vectorised check of the result: ascending lengths equal to
the sorted input lengths, offsets = padded prefix sums, every residue accounted for (per-letter histogram against the input
text), partition counts, stability (equal lengths keep input order) on a sample, and the DB loads (dbinspect).
    python tools/makedb_scale.py [N] [mem] [workdir]"""
import os, subprocess, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 25_000_000
MEM = sys.argv[2] if len(sys.argv) > 2 else "512M"
WORK = sys.argv[3] if len(sys.argv) > 3 else "/tmp/makedb_scale"
os.makedirs(WORK, exist_ok=True)
fasta, prefix = os.path.join(WORK, "big.fa"), os.path.join(WORK, "db")
LETTERS = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVXBZ", dtype=np.uint8)
rng = np.random.default_rng(5)
t0 = time.time()
lengths_in = np.clip(np.exp(rng.normal(np.log(40.0), 0.7, N)).astype(np.int64), 1, 20000)
lengths_in[:5] = [20000, 9000, 8001, 8000, 1281]
hist_in = np.zeros(256, dtype=np.int64)
with open(fasta, "wb") as f:
    step = 500_000
    for b in range(0, N, step):
        ls = lengths_in[b:b + step]
        seq = LETTERS[rng.integers(0, len(LETTERS), int(ls.sum()))]
        hist_in += np.bincount(seq, minlength=256)
        ends = np.cumsum(ls)
        parts = []
        pos = 0
        sb = seq.tobytes()
        for i, e in enumerate(ends.tolist()):
            parts.append(b">s%d\n" % (b + i))
            parts.append(sb[pos:e])
            parts.append(b"\n")
            pos = e
        f.write(b"".join(parts))
print("FASTA: %d sequences, %d residues, %.2f GB, written in %.0f s" % (N, int(lengths_in.sum()), os.path.getsize(fasta) / 1e9, time.time() - t0), flush=True)
t0 = time.time()
exe = os.path.join(ROOT, "cudasw4_amd", "lib", "makedb")
p = subprocess.run([exe, fasta, prefix, "--mem", MEM, "--tempdir", WORK], capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="8"))
print(p.stdout.strip()); print("makedb --mem %s: rc %d, %.0f s wall" % (MEM, p.returncode, time.time() - t0), flush=True)
assert p.returncode == 0, p.stderr
assert "Memory limit reached" in p.stdout
lengths = np.fromfile(prefix + "0lengths", dtype=np.int32)
offsets = np.fromfile(prefix + "0offsets", dtype=np.uint64)
assert len(lengths) == N and (np.diff(lengths) >= 0).all() and (np.sort(lengths_in) == lengths).all()
padded = (lengths.astype(np.int64) + 3) // 4 * 4
assert offsets[0] == 0 and (np.diff(offsets.astype(np.int64)) == padded).all()
chars = np.memmap(prefix + "0chars", dtype=np.int8, mode="r")
assert len(chars) == int(offsets[-1])
hist_db = np.zeros(21, dtype=np.int64)
for b in range(0, len(chars), 1 << 28):
    hist_db += np.bincount(np.asarray(chars[b:b + (1 << 28)]).astype(np.int64), minlength=21)[:21]
table = O.encode(bytes(range(256)))
want = np.zeros(21, dtype=np.int64)
np.add.at(want, table.astype(np.int64), hist_in)
want[20] += int(padded.sum() - lengths.astype(np.int64).sum())
assert (hist_db == want).all(), (hist_db, want)
hoff = np.fromfile(prefix + "0headeroffsets", dtype=np.uint64)
hdr = np.memmap(prefix + "0headers", dtype=np.uint8, mode="r")
pick = np.unique(np.concatenate([np.arange(0, N, max(1, N // 20000)), np.arange(N - 100, N)]))
ids = np.array([int(bytes(hdr[int(hoff[i]) + 1:int(hoff[i + 1])])) for i in pick])
assert (lengths_in[ids] == lengths[pick]).all()
run = np.arange(N // 2, N // 2 + 20000)   # a run of (mostly) equal lengths: input order kept
rid = np.array([int(bytes(hdr[int(hoff[i]) + 1:int(hoff[i + 1])])) for i in run])
same = lengths[run][1:] == lengths[run][:-1]
assert (rid[1:][same] > rid[:-1][same]).all()
meta = open(prefix + "0metadata", "rb").read()
counts = np.frombuffer(meta[4 + 36 * 4:], dtype=np.uint64)
assert counts.sum() == N and counts.tolist() == np.histogram(lengths, bins=np.concatenate([[0], O.partition_boundaries().astype(np.int64) + 1]))[0].tolist()
info = subprocess.run([os.path.join(ROOT, "cudasw4_amd", "lib", "dbinspect"), prefix, "8"], capture_output=True, text=True)
assert info.returncode == 0, info.stderr
d = __import__("json").loads(info.stdout.strip().splitlines()[-1]); print("dbinspect: num_sequences %d, num_chars %d, residues %d, partition_counts %s" % (d["num_sequences"], d["num_chars"], d["residues"], d["partition_counts"]))
print("verified: lengths, offsets, letter histogram (%d letters), headers of %d sampled sequences, stability on a 20000-sequence run, partition counts, dbinspect" % (int(hist_db.sum()), len(pick)))
for f in os.listdir(WORK):
    os.remove(os.path.join(WORK, f))
