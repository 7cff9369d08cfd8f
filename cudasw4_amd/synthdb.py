"""Synthetic databases for benchmarks and tests (numpy only, seeded): stand-ins for the real FASTA files the
reference's benchmark scripts download (runsprotbenchmark.sh:18-51 etc. — there is no network on the build and GPU
boxes).  Output is in dbdata layout: int8 codes padded to a multiple of 4 with code 20,
uint64 offsets, int32 true lengths, ascending length like makedb writes it.

Two generators:
  random_db    independent residues (uniform, or any composition) for given lengths — no two sequences are related, so
               no query ever scores far above the noise floor and no packed kernel ever flags an overflow;
  sprot_like   the Swiss-Prot stand-in of BASELINE config 3: Swiss-Prot length histogram and amino-acid composition
               PLUS seeded protein families of the query set (mutated copies, indels, fragments in random flanks), so
               that a scan carries the load real data carries: hits far above the noise floor, overflow lists and
               32-bit re-score launches (half2_kernels.cuh:1087-1109, cudasw4.cuh:2134-2172).
"""
import os

import numpy as np

SPROT_SEQUENCES = 570_000      # UniProtKB/Swiss-Prot order of magnitude
SPROT_MAX_LENGTH = 35_213      # titin
TREMBL_SEQUENCES = 250_000_000   # UniProtKB/TrEMBL order of magnitude (runtremblbenchmark.sh: 57 GB gzipped FASTA): ~9.6e10 residues here
UNIREF50_SEQUENCES = 60_000_000  # UniRef50 order of magnitude (rununiref50benchmark.sh: 12 GB gzipped FASTA): ~2.3e10 residues here

# amino-acid composition of UniProtKB/Swiss-Prot (release statistics, per cent) in the code order of ConvertAA_20
# (convert.cuh:32: A R N D C Q E G H I L K M F P S T W Y V)
SPROT_COMPOSITION = np.array([8.25, 5.53, 4.06, 5.46, 1.38, 3.93, 6.72, 7.07, 2.27, 5.91,
                              9.65, 5.80, 2.41, 3.86, 4.74, 6.65, 5.36, 1.10, 2.92, 6.85])

_LETTERS = b"ARNDCQEGHILKMFPSTWYV"


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


def _layout(lengths):
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    padded = (lengths.astype(np.int64) + 3) // 4 * 4
    offsets = np.zeros(len(lengths) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(padded)
    return lengths, offsets


def _pad_with_other(chars, offsets, lengths):
    # padding bytes -> 20 (at most three per sequence)
    ends = offsets[:-1].astype(np.int64) + lengths.astype(np.int64)
    nxt = offsets[1:].astype(np.int64)
    for k in range(3):
        pos = ends + k
        chars[pos[pos < nxt]] = 20


def _residues(rng, total, composition):
    if composition is None:
        return rng.integers(0, 20, total, dtype=np.int8)
    # inverse CDF through a 65536-entry table: one 16-bit draw and one gather per residue
    p = np.asarray(composition, dtype=np.float64)
    edges = np.round(np.cumsum(p / p.sum()) * 65536.0).astype(np.int64)
    edges[-1] = 65536
    lut = np.repeat(np.arange(len(p), dtype=np.int8), np.diff(np.concatenate([[0], edges])))
    out = np.empty(total, dtype=np.int8)
    step = 1 << 25  # bounded temporaries for multi-GB databases
    for b in range(0, total, step):
        e = min(total, b + step)
        out[b:e] = lut[rng.integers(0, 65536, e - b, dtype=np.uint16)]
    return out


def random_db(lengths, seed=7, other_fraction=0.0, composition=None):
    """Independent random residues (codes 0..19, uniform or with the given composition; a fraction `other_fraction` of
    code 20 = unknown letters) for the given ascending lengths -> (chars, offsets, lengths)."""
    lengths, offsets = _layout(lengths)
    total = int(offsets[-1])
    rng = np.random.default_rng(seed)
    chars = _residues(rng, total, composition)
    if other_fraction > 0:
        chars[rng.random(total) < other_fraction] = 20
    _pad_with_other(chars, offsets, lengths)
    return chars, offsets, lengths


# ---------------------------------------------------------------------------------------------- protein families
def default_family_seeds():
    """The 20 benchmark queries (tests/golden/allqueries.fasta == the reference's allqueries.fasta) as code arrays:
    real UniProt proteins, whose relatives a real Swiss-Prot holds."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "allqueries.fasta")
    lut = np.full(256, 20, dtype=np.int8)
    for i, c in enumerate(_LETTERS):
        lut[c] = i
    seqs, cur = [], []
    with open(path, "rb") as f:
        for line in f:
            line = line.strip()
            if line.startswith(b">"):
                if cur:
                    seqs.append(lut[np.frombuffer(b"".join(cur), dtype=np.uint8)])
                cur = []
            elif line:
                cur.append(line)
    if cur:
        seqs.append(lut[np.frombuffer(b"".join(cur), dtype=np.uint8)])
    return seqs


def mutate(rng, seq, identity, indel_rate=0.01, composition=None):
    """A relative of `seq`: every residue is replaced with probability 1 - identity (by a residue drawn from the
    composition), and about indel_rate * len insertions / deletions of 1..30 residues are applied."""
    s = np.array(seq, dtype=np.int8, copy=True)
    n = len(s)
    sub = rng.random(n) >= identity
    s[sub] = _residues(rng, int(sub.sum()), composition)
    nindel = rng.poisson(max(indel_rate * n, 0.0))
    for _ in range(int(nindel)):
        if len(s) < 8:
            break
        pos = int(rng.integers(0, len(s)))
        k = int(min(rng.integers(1, 31), len(s) // 4))
        if rng.random() < 0.5:
            s = np.concatenate([s[:pos], s[pos + k:]])
        else:
            s = np.concatenate([s[:pos], _residues(rng, k, composition), s[pos:]])
    return s


def family_members(seeds, seed=2031, min_size=50, max_size=500, composition=SPROT_COMPOSITION, max_len=SPROT_MAX_LENGTH,
                   size_caps=None, with_flags=False):
    """For every seed protein a family of min_size..max_size relatives (seeded; at most size_caps[i] for seed i):
      * the protein itself (real Swiss-Prot holds the benchmark queries);
      * 60 %: full-length relatives, 40..99 % identity, indels;
      * 25 %: fragments (30..100 % of the protein), 50..95 % identity;
      * 15 %: a domain of the protein (20..60 %) inside unrelated flanks of 0.2..2x its length — multi-domain relatives,
              which put hits into OTHER length classes than the query's own.
    -> list of code arrays (with_flags: and a list that marks the proteins themselves)."""
    rng = np.random.default_rng(seed)
    out, flags = [], []
    for si, s in enumerate(seeds):
        n = len(s)
        size = int(rng.integers(min_size, max_size + 1))
        if size_caps is not None:
            size = max(1, min(size, int(size_caps[si])))
        out.append(np.array(s[:max_len], dtype=np.int8, copy=True))
        flags.append(True)
        for _ in range(size - 1):
            u = rng.random()
            if u < 0.60:
                m = mutate(rng, s, identity=rng.uniform(0.40, 0.99), composition=composition)
            elif u < 0.85:
                frac = rng.uniform(0.3, 1.0)
                k = max(8, int(n * frac))
                b = int(rng.integers(0, n - k + 1))
                m = mutate(rng, s[b:b + k], identity=rng.uniform(0.50, 0.95), composition=composition)
            else:
                frac = rng.uniform(0.2, 0.6)
                k = max(8, int(n * frac))
                b = int(rng.integers(0, n - k + 1))
                core = mutate(rng, s[b:b + k], identity=rng.uniform(0.50, 0.95), composition=composition)
                fl = int(k * rng.uniform(0.2, 2.0))
                left = int(rng.integers(0, fl + 1))
                m = np.concatenate([_residues(rng, left, composition), core, _residues(rng, fl - left, composition)])
            out.append(m[:max_len])
            flags.append(False)
    return (out, flags) if with_flags else out


def sprot_like(n=SPROT_SEQUENCES, seed=2024, max_len=SPROT_MAX_LENGTH, families=True, seeds=None, return_family_ids=False):
    """The Swiss-Prot-like DB of BASELINE config 3 -> (chars, offsets, lengths).

    Background: n sequences with Swiss-Prot's length histogram (sprot_like_lengths) and amino-acid composition.
    families=True (default): the relatives of the seed proteins (default: the 20 benchmark queries; 50..500 per protein
    for n >= 100 000, scaled down below, and never more than a quarter of the background sequences of comparable
    length — Swiss-Prot has only a few hundred proteins above 4000 residues) REPLACE background sequences of the nearest
    length at or above their own, so that the length histogram, the sequence count and the sort order stay those of the
    background (a longer slot keeps its random tail behind the relative).  families=False: independent residues only
    (the round-1..3 stand-in, now with the Swiss-Prot composition).
    return_family_ids: also return the sorted positions of the family members."""
    lengths = sprot_like_lengths(n, seed, max_len)
    chars, offsets, lengths = random_db(lengths, seed=seed + 7, composition=SPROT_COMPOSITION)
    fam_pos = np.zeros(0, dtype=np.int64)
    if families:
        if seeds is None:
            seeds = default_family_seeds()
        scale = min(1.0, n / 100_000.0)
        # slots at or above a seed's own length (up to 1.6x) that its family may take: a third of them, shared by the
        # seeds whose windows overlap
        def window(s):
            return (int(np.searchsorted(lengths, len(s), side="left")), int(np.searchsorted(lengths, int(1.6 * len(s)), side="right")))
        caps = []
        for s in seeds:
            lo, hi = window(s)
            sharers = sum(1 for t in seeds if window(t)[0] < hi and lo < window(t)[1])
            caps.append(max(1, (hi - lo) // (3 * max(sharers, 1))))
        members, is_self = family_members(seeds, seed=seed + 13, min_size=max(2, int(50 * scale)),
                                          max_size=max(4, int(500 * scale)), max_len=max_len, size_caps=caps, with_flags=True)
        # the proteins themselves first, then the relatives, longest first: what is scarce (long slots) goes to those
        # that need it
        order = sorted(range(len(members)), key=lambda i: (not is_self[i], -len(members[i]), i))
        rng = np.random.default_rng(seed + 17)
        # next free slot at or above i (path-compressed), n = none
        nxt = np.arange(n + 1, dtype=np.int64)

        def free_at_or_above(i):
            r = i
            while nxt[r] != r:
                r = int(nxt[r])
            while nxt[i] != r:
                nxt[i], i = r, int(nxt[i])
            return r
        fam = []
        for mi in order:
            m = members[mi]
            want = len(m)
            j = free_at_or_above(min(int(np.searchsorted(lengths, want, side="left")), n))
            if j >= n:
                # nothing free at or above: the longest free slot below holds a window of the relative
                j = n - 1
                while j >= 0 and nxt[j] != j:
                    j -= 1
                if j < 0:
                    break
            nxt[j] = j + 1
            L = int(lengths[j])
            o = int(offsets[j])
            k = min(L, want)
            b = int(rng.integers(0, want - k + 1))
            chars[o:o + k] = m[b:b + k]
            fam.append(j)
        fam_pos = np.array(sorted(fam), dtype=np.int64)
    if return_family_ids:
        return chars, offsets, lengths, fam_pos
    return chars, offsets, lengths


def uniref50_like(n=UNIREF50_SEQUENCES, seed=50, max_len=SPROT_MAX_LENGTH, torch_device=None):
    """A UniRef50-sized stand-in (BASELINE configs 4 and 5: rununiref50benchmark.sh / runtremblbenchmark.sh download 12 /
    57 GB of FASTA): n sequences with the Swiss-Prot-like length histogram and composition, independent residues (UniRef50
    is clustered at 50 % identity: close relatives are rare by construction).  ~385 residues per sequence: the default
    6e7 sequences are 2.3e10 residues, 23 GB of chars.  torch_device: draw the residues on that GPU (numpy needs ~20 s
    per 10^9 residues on one core) -> (chars, offsets, lengths) as numpy arrays in dbdata layout."""
    lengths, offsets = _layout(sprot_like_lengths(n, seed, max_len))
    total = int(offsets[-1])
    if torch_device is None:
        chars = _residues(np.random.default_rng(seed + 7), total, SPROT_COMPOSITION)
    else:
        import torch
        p = np.asarray(SPROT_COMPOSITION, dtype=np.float64)
        edges = np.round(np.cumsum(p / p.sum()) * 65536.0).astype(np.int64)
        edges[-1] = 65536
        lut = torch.from_numpy(np.repeat(np.arange(20, dtype=np.int8), np.diff(np.concatenate([[0], edges])))).to(torch_device)
        gen = torch.Generator(device=torch_device)
        gen.manual_seed(seed + 7)
        chars = np.empty(total, dtype=np.int8)
        step = 1 << 29
        for b in range(0, total, step):
            e = min(total, b + step)
            draw = torch.randint(0, 65536, (e - b,), dtype=torch.int32, device=torch_device, generator=gen)
            chars[b:e] = lut[draw.long()].cpu().numpy()
            del draw
    _pad_with_other(chars, offsets, lengths)
    return chars, offsets, lengths
