"""ctypes binding of include/cudasw4_amd_driver.h (libcudasw4_host.so): the C++ host driver.

Plumbing only — the library drives libcudasw4_amd.so; there is no CPU path.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUDASW4_AMD_HOST_LIB: another build of the same C ABI (tests/host/fake_gpu: the real driver on a two-device fake runtime)
LIB_PATH = os.environ.get("CUDASW4_AMD_HOST_LIB") or os.path.join(_HERE, "lib", "libcudasw4_host.so")
MAKEDB = os.path.join(_HERE, "lib", "makedb")
ALIGN = os.path.join(_HERE, "lib", "align")

EXPORTS = ["swdrv_last_error", "swdrv_create", "swdrv_destroy", "swdrv_open_db", "swdrv_pseudo_db", "swdrv_upload",
           "swdrv_num_sequences", "swdrv_num_gpus", "swdrv_set_num_top", "swdrv_scan", "swdrv_reference_length",
           "swdrv_reference_header", "swdrv_encode", "swdrv_pseudo_sequence", "swdrv_matrix", "swdrv_reader_open",
           "swdrv_reader_next", "swdrv_reader_close", "swdrv_db_from_arrays", "swdrv_set_shard",
           "swdrv_record_kernel_events", "swdrv_take_kernel_events", "swdrv_shard_info", "swdrv_last_scores",
           "swdrv_batch_intervals", "swdrv_gpu_spans", "swdrv_plan_runs", "swdrv_shard_ranges", "swdrv_matrix25",
           "swdrv_encode25", "swdrv_last_rescored", "swdrv_scan_submit", "swdrv_scan_collect", "swdrv_in_flight",
           "swdrv_cached_chars", "swdrv_streamed_bytes", "swdrv_plan_residency", "swdrv_numa_node", "swdrv_device_of",
           "swdrv_bind_to_numa_node", "swdrv_device_numa_node", "swdrv_window_stats", "swdrv_service_launches",
           "swdrv_tail_overlaps", "swdrv_prefers_two_in_flight", "swdrv_pipeline_launches", "swdrv_handshake_active", "swdrv_preferred_in_flight",
           "swdrv_latency_scans", "swdrv_plan_runs_mode"]


class DriverError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcudasw4_host.so is not built: run __graft_entry__.build()")
    # PyTorch wheels bundle their own HIP runtime.  If this library pulled in the system's libamdhip64 first, a later
    # `import torch` would bring a second runtime into the process and find no GPU: load torch's first when it is there.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
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
    L.swdrv_scan_submit.argtypes = [vp, ctypes.c_char_p, i32]
    L.swdrv_scan_collect.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.swdrv_in_flight.argtypes = [vp]
    L.swdrv_cached_chars.restype = ctypes.c_int64
    L.swdrv_cached_chars.argtypes = [vp, ctypes.c_int]
    L.swdrv_streamed_bytes.restype = ctypes.c_int64
    L.swdrv_streamed_bytes.argtypes = [vp]
    L.swdrv_plan_residency.argtypes = [vp, sz, i32, sz, sz, sz, sz, sz, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), vp, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_int64)]
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
    i64 = ctypes.c_int64
    L.swdrv_db_from_arrays.argtypes = [vp, vp, sz, vp, vp, sz]
    L.swdrv_set_shard.argtypes = [vp, ctypes.c_int, ctypes.c_int, i64]
    L.swdrv_record_kernel_events.argtypes = [vp, ctypes.c_int]
    L.swdrv_take_kernel_events.argtypes = [vp, vp, ctypes.c_int]
    L.swdrv_shard_info.argtypes = [vp, ctypes.c_int, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64),
                                   ctypes.POINTER(ctypes.c_int)]
    L.swdrv_last_scores.argtypes = [vp, ctypes.c_int, vp, vp]
    L.swdrv_batch_intervals.argtypes = [vp, vp, ctypes.c_int]
    L.swdrv_gpu_spans.argtypes = [vp, vp, ctypes.c_int]
    L.swdrv_plan_runs.argtypes = [vp, sz, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int]
    L.swdrv_plan_runs_mode.argtypes = [vp, sz, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int]
    L.swdrv_shard_ranges.argtypes = [vp, vp, sz, ctypes.c_int, vp]
    L.swdrv_matrix25.argtypes = [ctypes.c_int, vp]
    L.swdrv_last_rescored.argtypes = [vp]
    L.swdrv_numa_node.argtypes = [vp, ctypes.c_int]
    L.swdrv_service_launches.restype = ctypes.c_int64
    L.swdrv_service_launches.argtypes = [vp]
    L.swdrv_latency_scans.restype = ctypes.c_int64
    L.swdrv_latency_scans.argtypes = [vp]
    L.swdrv_pipeline_launches.restype = ctypes.c_int64
    L.swdrv_pipeline_launches.argtypes = [vp]
    L.swdrv_preferred_in_flight.restype = ctypes.c_int
    L.swdrv_preferred_in_flight.argtypes = [vp, ctypes.c_int32]
    L.swdrv_handshake_active.restype = ctypes.c_int
    L.swdrv_handshake_active.argtypes = [vp]
    L.swdrv_tail_overlaps.restype = ctypes.c_int64
    L.swdrv_tail_overlaps.argtypes = [vp]
    L.swdrv_prefers_two_in_flight.argtypes = [vp, ctypes.c_int32]
    L.swdrv_window_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    L.swdrv_device_of.argtypes = [vp, ctypes.c_int]
    L.swdrv_bind_to_numa_node.argtypes = [ctypes.c_int]
    L.swdrv_device_numa_node.argtypes = [ctypes.c_int]
    L.swdrv_encode25.argtypes = [ctypes.c_char_p, vp, sz]
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


def matrix25(which=62) -> np.ndarray:
    """The full 25 x 25 table, letter order ARNDCQEGHILKMFPSTWYVBJZX*."""
    out = np.empty(625, dtype=np.int8)
    if lib.swdrv_matrix25(which, out.ctypes.data) != 0:
        raise ValueError("unknown matrix %r" % which)
    return out


def encode25(letters) -> np.ndarray:
    """Query letters -> codes 0..24 for the 25-letter tables (anything else -> X)."""
    if isinstance(letters, str):
        letters = letters.encode()
    out = np.empty(len(letters), dtype=np.int8)
    lib.swdrv_encode25(letters, out.ctypes.data, len(letters))
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


def device_numa_node(device):
    """NUMA node of HIP device `device` (its PCI function's numa_node attribute; -1: unknown)."""
    return int(lib.swdrv_device_numa_node(int(device)))


def bind_to_numa_node(node):
    """Run the calling thread (and the threads it starts) on the CPUs of `node` that the process may use -> True if bound."""
    return lib.swdrv_bind_to_numa_node(int(node)) == 0


def _check(rc):
    if rc != 0:
        raise DriverError(lib.swdrv_last_error().decode())


def plan_runs(sorted_lengths, kind_single, kind_many_small, kind_many_large, latency_mode=False):
    """The launch planner of the C++ driver (plan_launch_runs; no GPU needed): runs of a length-sorted subject list,
    largest partition first -> list of dicts (kind, part_id, begin, end, maxlen).  latency_mode: the plan of small shards
    of real DBs (partition 34 keeps a launch of its own whatever its size)."""
    l = np.ascontiguousarray(sorted_lengths, dtype=np.int32)
    out = np.zeros(36 * 5, dtype=np.int64)
    n = lib.swdrv_plan_runs_mode(l.ctypes.data, len(l), kind_single, kind_many_small, kind_many_large, int(bool(latency_mode)),
                                 out.ctypes.data, 36)
    if n < 0:
        raise DriverError(lib.swdrv_last_error().decode())
    return [{"kind": int(out[5 * i]), "part_id": int(out[5 * i + 1]), "begin": int(out[5 * i + 2]),
             "end": int(out[5 * i + 3]), "maxlen": int(out[5 * i + 4])} for i in range(n)]


def shard_ranges(offsets, sorted_lengths, world):
    """partitionDBAmongstGpus as the C++ driver does it (shard_database; no GPU needed):
    ranges[rank][partition] = (begin, end)."""
    l = np.ascontiguousarray(sorted_lengths, dtype=np.int32)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    out = np.zeros(world * 36 * 2, dtype=np.int64)
    _check(lib.swdrv_shard_ranges(l.ctypes.data, o.ctypes.data, len(l), world, out.ctypes.data))
    out = out.reshape(world, 36, 2)
    return [[(int(b), int(e)) for b, e in out[r]] for r in range(world)]


def plan_residency(local_offsets, max_len, max_gpu_mem=0, max_batch_bytes=0, max_batch_sequences=0, max_temp_bytes=0,
                   free_mem=288 << 30, allow_cache=True):
    """The C++ driver's residency decision for one GPU's shard (plan_residency; no GPU needed) -> dict: cache_begin (the
    subjects from there on stay in device memory), cache_bytes, batch_bytes, batches = [(begin, end), ...] of the rest,
    temp_per_stream (cap of each of the 4 stripe-border scratch buffers)."""
    o = np.ascontiguousarray(local_offsets, dtype=np.uint64)
    n = len(o) - 1
    cap = max(n, 1)
    out = np.zeros(2 * cap, dtype=np.int64)
    cb, cby, bb, tps = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    nb = lib.swdrv_plan_residency(o.ctypes.data, n, int(max_len), max_gpu_mem, max_batch_bytes, max_batch_sequences,
                                  max_temp_bytes, free_mem, int(allow_cache), ctypes.byref(cb), ctypes.byref(cby),
                                  ctypes.byref(bb), out.ctypes.data, cap, ctypes.byref(tps))
    if nb < 0:
        raise DriverError(lib.swdrv_last_error().decode())
    return {"cache_begin": cb.value, "cache_bytes": cby.value, "batch_bytes": bb.value, "temp_per_stream": tps.value,
            "batches": [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(nb)]}


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

    def db_from_arrays(self, chars, offsets, lengths):
        """DB from numpy arrays in dbdata layout (sorted by ascending length); the driver copies them."""
        c = np.ascontiguousarray(chars, dtype=np.int8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        l = np.ascontiguousarray(lengths, dtype=np.int32)
        _check(lib.swdrv_db_from_arrays(self.handle, c.ctypes.data, len(c), o.ctypes.data, l.ctypes.data, len(l)))

    def set_shard(self, rank, world, id_base=0):
        """One process per GPU: take shard `rank` of `world` (call before loading the DB)."""
        _check(lib.swdrv_set_shard(self.handle, rank, world, id_base))

    def upload(self):
        _check(lib.swdrv_upload(self.handle))

    def record_kernel_events(self, on=True):
        _check(lib.swdrv_record_kernel_events(self.handle, int(on)))

    def take_kernel_events(self):
        """-> list of dicts (gpu, kind, part_id, qlen, subjects, cells, chars, ms, t0_ms, t1_ms, eff_kind, rows,
        nstripes, lanes, rescore), HIP-event timed launches; t0/t1: begin and end on the device clock since recording was
        switched on; the last four name the kernel instantiation."""
        cap = 16384
        while True:
            buf = np.zeros(cap * 15, dtype=np.float64)
            n = lib.swdrv_take_kernel_events(self.handle, buf.ctypes.data, cap)
            if n < 0:
                raise DriverError(lib.swdrv_last_error().decode())
            if n <= cap:
                break
            raise DriverError("more than %d kernel events between two takes" % cap)
        keys = ("gpu", "kind", "part_id", "qlen", "subjects", "cells", "chars", "ms", "t0_ms", "t1_ms", "eff_kind", "rows",
                "nstripes", "lanes", "rescore")
        floats = ("cells", "chars", "ms", "t0_ms", "t1_ms")
        return [dict(zip(keys, (float(v) if k in floats else int(v) for k, v in zip(keys, buf[15 * i:15 * i + 15]))))
                for i in range(n)]

    def shard_info(self, gpu=0):
        i64 = ctypes.c_int64
        n, r, c, res = i64(), i64(), i64(), ctypes.c_int()
        _check(lib.swdrv_shard_info(self.handle, gpu, ctypes.byref(n), ctypes.byref(r), ctypes.byref(c), ctypes.byref(res)))
        return {"subjects": n.value, "residues": r.value, "chars": c.value, "resident": bool(res.value),
                "cached_chars": int(lib.swdrv_cached_chars(self.handle, gpu))}

    def window_stats(self):
        """(side launches that scanned long subjects as overlapping windows, windows scanned) since the driver was created."""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        _check(lib.swdrv_window_stats(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def service_launches(self):
        """Re-score service launches since the driver was created."""
        return int(lib.swdrv_service_launches(self.handle))

    def prefers_two_in_flight(self, query_length=0):
        """True when the tail hand-over applies (small resident shards, or a query of query_length residues that is scanned
        in a few milliseconds): use scan_many / submit + collect."""
        return bool(lib.swdrv_prefers_two_in_flight(self.handle, int(query_length)))

    def latency_scans(self):
        """Scans planned in latency mode (swdrv_latency_scans)."""
        return int(lib.swdrv_latency_scans(self.handle))

    def handshake_active(self):
        """True when the start handshake passed its probe (swdrv_handshake_active)."""
        return int(lib.swdrv_handshake_active(self.handle)) == 1

    def pipeline_launches(self):
        """Side launches of the longest subjects that ran as pipelines of one-wave stages over many CUs (swdrv_pipeline_launches)."""
        return int(lib.swdrv_pipeline_launches(self.handle))

    def tail_overlaps(self):
        """Queries whose bulk launch was gated on the dry signal of the query before (swdrv_tail_overlaps)."""
        return int(lib.swdrv_tail_overlaps(self.handle))

    def numa_node(self, gpu=0):
        """NUMA node of the GPU's PCI function (-1: unknown)."""
        return int(lib.swdrv_numa_node(self.handle, gpu))

    def device_of(self, gpu=0):
        return int(lib.swdrv_device_of(self.handle, gpu))

    def streamed_bytes(self):
        """Subject bytes copied host -> device by scans since the driver was created (all GPUs)."""
        return int(lib.swdrv_streamed_bytes(self.handle))

    def last_scores(self, gpu=0):
        """Every score of the last scan on one GPU and the global id of each position."""
        n = self.shard_info(gpu)["subjects"]
        s = np.zeros(max(n, 1), dtype=np.float32)
        i = np.zeros(max(n, 1), dtype=np.int64)
        _check(lib.swdrv_last_scores(self.handle, gpu, s.ctypes.data, i.ctypes.data))
        return s[:n].astype(np.int32), i[:n]

    def all_scores(self):
        """Scores of the last scan for every subject of this driver's shards, indexed by global id -> (ids, scores)."""
        ids, sc = [], []
        for g in range(self.num_gpus()):
            s, i = self.last_scores(g)
            ids.append(i)
            sc.append(s)
        return np.concatenate(ids), np.concatenate(sc)

    def batch_intervals(self):
        buf = np.zeros(3 * 65536, dtype=np.float32)
        n = lib.swdrv_batch_intervals(self.handle, buf.ctypes.data, 65536)
        if n < 0:
            raise DriverError(lib.swdrv_last_error().decode())
        n = min(n, 65536)
        return [(int(buf[3 * i]), float(buf[3 * i + 1]), float(buf[3 * i + 2])) for i in range(n)]

    def gpu_spans(self):
        buf = np.zeros(2 * 64, dtype=np.float64)
        n = lib.swdrv_gpu_spans(self.handle, buf.ctypes.data, 64)
        if n < 0:
            raise DriverError(lib.swdrv_last_error().decode())
        return [(float(buf[2 * i]), float(buf[2 * i + 1])) for i in range(n)]

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
                "num_rescored": int(lib.swdrv_last_rescored(self.handle)),
                "seconds": sec.value, "gcups": gcups.value}

    def submit(self, query_letters):
        """First half of scan(): enqueue the query on every GPU without waiting (at most two in flight)."""
        if isinstance(query_letters, str):
            query_letters = query_letters.encode()
        _check(lib.swdrv_scan_submit(self.handle, query_letters, len(query_letters)))

    def collect(self):
        """Second half of scan(): the merged results of the oldest submitted query."""
        cap = max(self.num_top, 1)
        scores = np.zeros(cap, dtype=np.int32)
        ids = np.zeros(cap, dtype=np.int64)
        nres, novf = ctypes.c_int(), ctypes.c_int()
        sec, gcups = ctypes.c_double(), ctypes.c_double()
        _check(lib.swdrv_scan_collect(self.handle, scores.ctypes.data, ids.ctypes.data, cap, ctypes.byref(nres),
                                      ctypes.byref(novf), ctypes.byref(sec), ctypes.byref(gcups)))
        n = nres.value
        return {"scores": scores[:n].copy(), "ids": ids[:n].copy(), "num_overflows": novf.value,
                "num_rescored": int(lib.swdrv_last_rescored(self.handle)),
                "seconds": sec.value, "gcups": gcups.value}

    def scan_many(self, queries):
        """Every query of a list, pipelined: the next query is submitted before the current one is collected, so the
        GPUs stay busy across query boundaries (what `align` does with a query file).  -> list of result dicts."""
        out = []
        for q in queries:
            self.submit(q)
            while lib.swdrv_in_flight(self.handle) >= max(2, int(lib.swdrv_preferred_in_flight(self.handle, len(q)))):
                out.append(self.collect())
        while lib.swdrv_in_flight(self.handle) > 0:
            out.append(self.collect())
        return out

    def scan_stream(self, queries):
        """Every query of a list the way `align` processes a query file (align_main.cpp: processQueryFile): one query at a
        time like the reference, except where the driver says the tail hand-over applies to the query just submitted (a
        small resident shard, or a query that is scanned in a few milliseconds) — then the next query is submitted before
        that one is collected.  -> list of result dicts, in the order of the queries."""
        out = []
        for q in queries:
            self.submit(q)
            limit = int(lib.swdrv_preferred_in_flight(self.handle, len(q)))
            while lib.swdrv_in_flight(self.handle) >= limit:
                out.append(self.collect())
        while lib.swdrv_in_flight(self.handle) > 0:
            out.append(self.collect())
        return out

    def reference_length(self, i):
        return int(lib.swdrv_reference_length(self.handle, i))

    def reference_header(self, i):
        buf = ctypes.create_string_buffer(4096)
        lib.swdrv_reference_header(self.handle, i, buf, 4096)
        return buf.value.decode()
