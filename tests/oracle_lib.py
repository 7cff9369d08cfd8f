"""ctypes access to the CPU oracle (oracle/libsw_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import json
import os
import subprocess

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

_lib = None


def _cpu_has_avx512bw():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " avx512bw " in (line + " ")
    except OSError:
        pass
    return False


def build_oracle():
    """Compile the oracle with gcc if it is missing or stale; returns the library for this host's ISA level
    (AVX-512 build when the CPU has it, AVX2 build otherwise — no -march=native: the files travel)."""
    name = "libsw_oracle_v4.so" if _cpu_has_avx512bw() else "libsw_oracle.so"
    so = os.path.join(ORACLE_DIR, name)
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("sw_oracle.c", "sw_oracle.h", "blosum_tables.inc", "Makefile")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, name], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build_oracle())
        i8p = ctypes.POINTER(ctypes.c_int8)
        L.swo_blosum21.restype = i8p
        L.swo_blosum21.argtypes = [ctypes.c_int]
        L.swo_score.restype = ctypes.c_int32
        L.swo_score.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
        for name in ("swo_scan", "swo_scan_simd", "swo_scan_striped"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                          ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                          ctypes.c_int]
        L.swo_encode.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
        L.swo_pseudodb_codes.argtypes = [ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p]
        L.swo_partition_boundaries.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.swo_partition_of.argtypes = [ctypes.c_int32]
        L.swo_topk.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def blosum21(which=62):
    p = lib().swo_blosum21(which)
    if not p:
        raise ValueError("unknown matrix %r" % which)
    return np.ctypeslib.as_array(p, shape=(441,)).astype(np.int8).copy()


def encode(seq) -> np.ndarray:
    if isinstance(seq, str):
        seq = seq.encode()
    out = np.empty(len(seq), dtype=np.int8)
    lib().swo_encode(seq, out.ctypes.data, len(seq))
    return out


def score(q, s, m21=None, gop=-11, gex=-1) -> int:
    q = np.ascontiguousarray(q, dtype=np.int8)
    s = np.ascontiguousarray(s, dtype=np.int8)
    m = blosum21(62) if m21 is None else np.ascontiguousarray(m21, dtype=np.int8)
    return int(lib().swo_score(q.ctypes.data, len(q), s.ctypes.data, len(s), m.ctypes.data, gop, gex))


def scan(q, chars, offsets, lengths, m21=None, gop=-11, gex=-1, simd=False, nthreads=0, striped=False) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.int8)
    chars = np.ascontiguousarray(chars, dtype=np.int8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    m = blosum21(62) if m21 is None else np.ascontiguousarray(m21, dtype=np.int8)
    out = np.empty(len(lengths), dtype=np.int32)
    f = lib().swo_scan_striped if striped else (lib().swo_scan_simd if simd else lib().swo_scan)
    f(q.ctypes.data, len(q), chars.ctypes.data, offsets.ctypes.data, lengths.ctypes.data, len(lengths),
      m.ctypes.data, gop, gex, out.ctypes.data, nthreads)
    return out


def pseudodb_codes(length, seed=42) -> np.ndarray:
    out = np.empty(length, dtype=np.int8)
    lib().swo_pseudodb_codes(length, seed, out.ctypes.data)
    return out


def partition_boundaries() -> np.ndarray:
    out = np.zeros(64, dtype=np.int32)
    n = lib().swo_partition_boundaries(out.ctypes.data, 64)
    return out[:n].copy()


def topk(scores, k):
    scores = np.ascontiguousarray(scores, dtype=np.int32)
    os_ = np.empty(k, dtype=np.int32)
    oi = np.empty(k, dtype=np.int64)
    lib().swo_topk(scores.ctypes.data, len(scores), k, os_.ctypes.data, oi.ctypes.data)
    return os_, oi


def max_threads() -> int:
    return int(lib().swo_max_threads())


# ---------------------------------------------------------------- helpers shared by tests / bench

def read_fasta(path):
    """Minimal FASTA reader for fixtures (headers, sequences as bytes)."""
    headers, seqs, cur = [], [], None
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                headers.append(line[1:].decode())
                cur = bytearray()
                seqs.append(cur)
            elif cur is not None:
                cur.extend(line)
    return headers, [bytes(s) for s in seqs]


def load_queries():
    headers, seqs = read_fasta(os.path.join(GOLDEN_DIR, "allqueries.fasta"))
    return headers, [encode(s) for s in seqs]


def make_db(seqs_encoded):
    """dbdata layout (Appendix C of SURVEY.md): chars padded with code 20 to a multiple of 4,
    uint64 offsets[N+1], int32 lengths[N].  Input order is kept (callers sort if they want)."""
    lengths = np.array([len(s) for s in seqs_encoded], dtype=np.int32)
    padded = (lengths.astype(np.int64) + 3) // 4 * 4
    offsets = np.zeros(len(seqs_encoded) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(padded)
    chars = np.full(int(offsets[-1]), 20, dtype=np.int8)
    for i, s in enumerate(seqs_encoded):
        chars[int(offsets[i]):int(offsets[i]) + len(s)] = s
    return chars, offsets, lengths


def golden(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)
