"""Synthetic databases for benchmarks and tests (numpy only, seeded): stand-ins for the real FASTA files the
reference's benchmark scripts download (runsprotbenchmark.sh:18-51 etc. — there is no network on the build and GPU
boxes).  Output is in dbdata layout: int8 codes padded to a multiple of 4 with code 20,
uint64 offsets, int32 true lengths, ascending length like makedb writes it."""
import numpy as np

SPROT_SEQUENCES = 570_000      # UniProtKB/Swiss-Prot order of magnitude
SPROT_MAX_LENGTH = 35_213      # titin


def sprot_like_lengths(n=SPROT_SEQUENCES, seed=2024, max_len=SPROT_MAX_LENGTH):
    """Log-normal lengths (median ~290, sigma 0.75) clipped to [2, max_len] with a few giant proteins (one per 30 000
    sequences, spread from 8000 to max_len) — the Swiss-Prot shape: every reference length partition is populated,
    partition 34 (1281..8000) holds ~2.4 % of the sequences and ~11 % of the residues, partition 35 a handful."""
    rng = np.random.default_rng(seed)
    l = np.exp(rng.normal(np.log(290.0), 0.75, n)).astype(np.int64)
    l = np.clip(l, 2, max_len)
    k = max(1, n // 30000)
    l[:k] = np.linspace(max_len, 8000, k).astype(np.int64)
    return np.sort(l).astype(np.int32)


def random_db(lengths, seed=7, other_fraction=0.0):
    """Uniform random residues (codes 0..19; a fraction `other_fraction` of code 20 = unknown letters) for the given
    ascending lengths -> (chars, offsets, lengths)."""
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    padded = (lengths.astype(np.int64) + 3) // 4 * 4
    offsets = np.zeros(len(lengths) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(padded)
    total = int(offsets[-1])
    rng = np.random.default_rng(seed)
    chars = rng.integers(0, 20, total, dtype=np.int8)
    if other_fraction > 0:
        chars[rng.random(total) < other_fraction] = 20
    # padding bytes -> 20 (at most three per sequence)
    ends = offsets[:-1].astype(np.int64) + lengths.astype(np.int64)
    nxt = offsets[1:].astype(np.int64)
    for k in range(3):
        pos = ends + k
        chars[pos[pos < nxt]] = 20
    return chars, offsets, lengths


def sprot_like(n=SPROT_SEQUENCES, seed=2024, max_len=SPROT_MAX_LENGTH):
    """The Swiss-Prot-like DB of BASELINE config 3 -> (chars, offsets, lengths)."""
    return random_db(sprot_like_lengths(n, seed, max_len), seed=seed + 7)
