"""ctypes binding of include/cudasw4_amd_driver.h (libcudasw4_host.so): the C++ host driver.

Plumbing only — the library drives libcudasw4_amd.so; there is no CPU path.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcudasw4_host.so")
MAKEDB = os.path.join(_HERE, "lib", "makedb")
ALIGN = os.path.join(_HERE, "lib", "align")

EXPORTS = ["swdrv_last_error", "swdrv_create", "swdrv_destroy", "swdrv_open_db", "swdrv_pseudo_db", "swdrv_upload",
           "swdrv_num_sequences", "swdrv_num_gpus", "swdrv_set_num_top", "swdrv_scan", "swdrv_reference_length",
           "swdrv_reference_header", "swdrv_encode", "swdrv_pseudo_sequence", "swdrv_matrix", "swdrv_reader_open",
           "swdrv_reader_next", "swdrv_reader_close"]


class DriverError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcudasw4_host.so is not built: run __graft_entry__.build()")
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32
    L.swdrv_last_error.restype = ctypes.c_char_p
    L.swdrv_create.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_int, sz, sz, sz, sz, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.POINTER(vp)]
    L.swdrv_destroy.argtypes = [vp]
    L.swdrv_open_db.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
    L.swdrv_pseudo_db.argtypes = [vp, sz, i32]
    L.swdrv_upload.argtypes = [vp]
    L.swdrv_num_sequences.restype = ctypes.c_int64
    L.swdrv_num_sequences.argtypes = [vp]
    L.swdrv_num_gpus.argtypes = [vp]
    L.swdrv_set_num_top.argtypes = [vp, ctypes.c_int]
    L.swdrv_scan.argtypes = [vp, ctypes.c_char_p, i32, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                             ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double),
                             ctypes.POINTER(ctypes.c_double)]
    L.swdrv_reference_length.restype = i32
    L.swdrv_reference_length.argtypes = [vp, ctypes.c_int64]
    L.swdrv_reference_header.argtypes = [vp, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int]
    L.swdrv_encode.argtypes = [ctypes.c_char_p, vp, sz]
    L.swdrv_pseudo_sequence.argtypes = [i32, ctypes.c_int, vp]
    L.swdrv_matrix.argtypes = [ctypes.c_int, vp]
    L.swdrv_reader_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.swdrv_reader_next.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz),
                                    ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz)]
    L.swdrv_reader_close.argtypes = [vp]
    return L


lib = _load()


# ---- input helpers of the host library (what `align` does to its inputs; no GPU needed) ----

def encode(letters) -> np.ndarray:
    """ConvertAA_20: residue letters -> int8 codes 0..20."""
    if isinstance(letters, str):
        letters = letters.encode()
    out = np.empty(len(letters), dtype=np.int8)
    lib.swdrv_encode(letters, out.ctypes.data, len(letters))
    return out


def pseudo_sequence(length, seed=42) -> np.ndarray:
    """The pseudo-DB subject (one mt19937(seed) sequence), as codes."""
    out = np.empty(length, dtype=np.int8)
    lib.swdrv_pseudo_sequence(length, seed, out.ctypes.data)
    return out


def matrix(which=62) -> np.ndarray:
    out = np.empty(441, dtype=np.int8)
    if lib.swdrv_matrix(which, out.ctypes.data) != 0:
        raise ValueError("unknown matrix %r" % which)
    return out


def read_sequences(path):
    """FASTA / FASTQ (.gz) -> (headers, sequences as bytes), parsed like the reference's kseq reader."""
    h = ctypes.c_void_p()
    _check(lib.swdrv_reader_open(path.encode(), ctypes.byref(h)))
    headers, seqs = [], []
    hp, sp = ctypes.c_char_p(), ctypes.c_char_p()
    hl, sl = ctypes.c_size_t(), ctypes.c_size_t()
    try:
        while lib.swdrv_reader_next(h, ctypes.byref(hp), ctypes.byref(hl), ctypes.byref(sp), ctypes.byref(sl)):
            headers.append(ctypes.string_at(hp, hl.value).decode())
            seqs.append(ctypes.string_at(sp, sl.value))
    finally:
        lib.swdrv_reader_close(h)
    return headers, seqs


def _check(rc):
    if rc != 0:
        raise DriverError(lib.swdrv_last_error().decode())


class Driver:
    """SearchDriver (C++) == the reference's CudaSW4 as `align` uses it."""

    def __init__(self, devices=None, num_top=10, matrix=62, kinds=(0, 0, 3, 3), max_gpu_mem=0, max_batch_bytes=0,
                 max_batch_sequences=0, max_temp_bytes=0, gop=-11, gex=-1, verbose=False):
        h = ctypes.c_void_p()
        if devices:
            arr = (ctypes.c_int * len(devices))(*devices)
            n = len(devices)
        else:
            arr, n = None, 0
        _check(lib.swdrv_create(arr, n, num_top, matrix, kinds[0], kinds[1], kinds[2], kinds[3], max_gpu_mem,
                                max_batch_bytes, max_batch_sequences, max_temp_bytes, gop, gex, int(verbose),
                                ctypes.byref(h)))
        self.handle = h
        self.num_top = num_top

    def close(self):
        if self.handle:
            lib.swdrv_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def open_db(self, prefix, prefetch=False):
        _check(lib.swdrv_open_db(self.handle, prefix.encode(), int(prefetch)))

    def pseudo_db(self, num, length):
        _check(lib.swdrv_pseudo_db(self.handle, num, length))

    def upload(self):
        _check(lib.swdrv_upload(self.handle))

    def num_sequences(self):
        return int(lib.swdrv_num_sequences(self.handle))

    def num_gpus(self):
        return int(lib.swdrv_num_gpus(self.handle))

    def set_num_top(self, k):
        lib.swdrv_set_num_top(self.handle, k)
        self.num_top = k

    def scan(self, query_letters):
        if isinstance(query_letters, str):
            query_letters = query_letters.encode()
        cap = max(self.num_top, 1)
        scores = np.zeros(cap, dtype=np.int32)
        ids = np.zeros(cap, dtype=np.int64)
        nres, novf = ctypes.c_int(), ctypes.c_int()
        sec, gcups = ctypes.c_double(), ctypes.c_double()
        _check(lib.swdrv_scan(self.handle, query_letters, len(query_letters), scores.ctypes.data, ids.ctypes.data, cap,
                              ctypes.byref(nres), ctypes.byref(novf), ctypes.byref(sec), ctypes.byref(gcups)))
        n = nres.value
        return {"scores": scores[:n].copy(), "ids": ids[:n].copy(), "num_overflows": novf.value,
                "seconds": sec.value, "gcups": gcups.value}

    def reference_length(self, i):
        return int(lib.swdrv_reference_length(self.handle, i))

    def reference_header(self, i):
        buf = ctypes.create_string_buffer(4096)
        lib.swdrv_reference_header(self.handle, i, buf, 4096)
        return buf.value.decode()
