"""Host-side mirror of the reference's search driver (class CudaSW4, cudasw4.cuh:496-839) on top of
the C ABI.  PyTorch is used only for device memory, streams and (in bench.py) torch.distributed.

    db = DeviceDB.from_arrays(chars, offsets, lengths, device=0)      # setDatabase + prefetchDBToGpus
    s = Searcher(device=0, num_top=10, matrix=blosum62_21x21)         # CudaSW4(...)
    s.set_database(db)
    res = s.scan(query_codes)                                         # CudaSW4::scan -> ScanResult

Every score is computed by libcudasw4_amd.so; nothing here computes alignments on the CPU.
"""
from dataclasses import dataclass, field

import numpy as np
import torch

from . import capi

# length_partitions.hpp:75-113 (restated; tests compare it with the reference's table)
PARTITION_BOUNDARIES = np.array(
    [48, 64] + list(range(80, 257, 16)) + list(range(288, 513, 32)) + list(range(576, 1281, 64)) + [8000, 2**31 - 2],
    dtype=np.int64)
NUM_PARTITIONS = len(PARTITION_BOUNDARIES)
assert NUM_PARTITIONS == 36


@dataclass
class KernelTypeConfig:
    """options.hpp:22-25 / cudasw4.cuh:841-855: which arithmetic kind handles which partitions."""
    single_pass: int = capi.KIND_F16X2       # partitions 0..33
    many_pass_small: int = capi.KIND_F16X2   # partition 34 (1281..8000)
    many_pass_large: int = capi.KIND_F32     # partition 35 (> 8000)
    overflow: int = capi.KIND_F32            # re-score of packed overflows

    @staticmethod
    def dpx():  # options.cpp:196-201 (--dpx)
        return KernelTypeConfig(capi.KIND_I16X2, capi.KIND_I16X2, capi.KIND_I32, capi.KIND_I32)

    def validate(self):
        if self.many_pass_small not in (capi.KIND_F16X2, capi.KIND_I16X2):
            raise ValueError("manyPassType_small must be Half2 or DPXs16 (cudasw4.cuh:847-849)")
        if self.many_pass_large not in (capi.KIND_F32, capi.KIND_I32):
            raise ValueError("manyPassType_large must be Float or DPXs32 (cudasw4.cuh:850-852)")
        if self.overflow not in (capi.KIND_F32, capi.KIND_I32):
            raise ValueError("overflowType must be Float or DPXs32 (cudasw4.cuh:853-855)")

    def kind_for_partition(self, part_id):
        if part_id < NUM_PARTITIONS - 2:
            return self.single_pass
        return self.many_pass_small if part_id == NUM_PARTITIONS - 2 else self.many_pass_large


@dataclass
class ScanResult:
    """cudasw4.cuh:82-86 ScanResult + BenchmarkStats."""
    scores: np.ndarray
    reference_ids: np.ndarray
    num_overflows: int = 0
    seconds: float = 0.0
    gcups: float = 0.0
    stats: dict = field(default_factory=dict)


def sort_db_by_length(chars, offsets, lengths):
    """makedb.cpp:188-195 sorts subjects by length; returns (chars, offsets, lengths, original_ids)."""
    lengths = np.asarray(lengths, dtype=np.int32)
    offsets = np.asarray(offsets, dtype=np.uint64)
    order = np.argsort(lengths, kind="stable")
    padded = (lengths.astype(np.int64) + 3) // 4 * 4
    new_off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    new_off[1:] = np.cumsum(padded[order])
    new_chars = np.full(int(new_off[-1]), 20, dtype=np.int8)
    chars = np.asarray(chars, dtype=np.int8)
    for k, i in enumerate(order):
        a = int(offsets[i] - offsets[0])
        new_chars[int(new_off[k]):int(new_off[k]) + int(lengths[i])] = chars[a:a + int(lengths[i])]
    return new_chars, new_off, lengths[order].copy(), order.astype(np.int64)


class DeviceDB:
    """A DB (shard) resident in HBM, in dbdata layout, sorted by ascending length.

    Mirrors GpuWorkingSet's cached-DB buffers (cudasw4.cuh:251-321) and the per-partition subject
    counts of computeTotalNumSequencePerLengthPartition (cudasw4.cuh:904-926)."""

    def __init__(self, chars_t, offsets_t, lengths_t, lengths_host, id_offset=0):
        self.chars = chars_t
        self.offsets = offsets_t
        self.lengths = lengths_t
        self.device = chars_t.device
        self.num_sequences = int(lengths_t.numel())
        self.id_offset = int(id_offset)
        lh = np.asarray(lengths_host, dtype=np.int64)
        if len(lh) > 1 and np.any(np.diff(lh) < 0):
            raise ValueError("DB must be sorted by ascending length (makedb does this)")
        self.lengths_host = lh
        self.total_residues = int(lh.sum())
        # partition i holds lengths in (b[i-1], b[i]]  (length_partitions.hpp:11)
        ends = np.searchsorted(lh, PARTITION_BOUNDARIES, side="right")
        self.part_begin = np.concatenate([[0], ends[:-1]]).astype(np.int64)
        self.part_end = ends.astype(np.int64)
        self.max_length = int(lh[-1]) if len(lh) else 0

    @staticmethod
    def from_arrays(chars, offsets, lengths, device=0, id_offset=0):
        dev = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        chars = np.ascontiguousarray(chars, dtype=np.int8)
        pad = (-len(chars)) % 16 + 16  # slack so 4-byte letter loads never leave the allocation
        chars_t = torch.empty(len(chars) + pad, dtype=torch.int8, device=dev)
        chars_t[:len(chars)].copy_(torch.from_numpy(chars))
        chars_t[len(chars):].fill_(20)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        offsets_t = torch.from_numpy(offsets.view(np.int64).copy()).to(dev)
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        lengths_t = torch.from_numpy(lengths.copy()).to(dev)
        return DeviceDB(chars_t, offsets_t, lengths_t, lengths, id_offset)

    @staticmethod
    def pseudo(num, length, codes, device=0):
        """PseudoDBdata (dbdata.hpp:222-272): `codes` (one sequence) replicated `num` times, built on
        the device.  `codes` comes from the caller (the C++ driver / oracle generator)."""
        dev = torch.device("cuda", device)
        stride = (length + 3) // 4 * 4
        one = torch.full((stride,), 20, dtype=torch.int8, device=dev)
        one[:length].copy_(torch.from_numpy(np.ascontiguousarray(codes[:length], dtype=np.int8)))
        chars_t = torch.empty(num * stride + 32, dtype=torch.int8, device=dev)
        chars_t[:num * stride].view(num, stride).copy_(one.unsqueeze(0).expand(num, stride))
        chars_t[num * stride:].fill_(20)
        offsets_t = torch.arange(num + 1, dtype=torch.int64, device=dev) * stride
        lengths_t = torch.full((num,), length, dtype=torch.int32, device=dev)
        db = DeviceDB.__new__(DeviceDB)
        db.chars, db.offsets, db.lengths, db.device = chars_t, offsets_t, lengths_t, dev
        db.num_sequences, db.id_offset = num, 0
        db.lengths_host = None
        db.total_residues = num * length
        pid = int(np.searchsorted(PARTITION_BOUNDARIES, length, side="left"))
        db.part_begin = np.zeros(NUM_PARTITIONS, dtype=np.int64)
        db.part_end = np.zeros(NUM_PARTITIONS, dtype=np.int64)
        db.part_begin[pid + 1:] = num
        db.part_end[pid:] = num
        db.max_length = length
        return db

    def partition_max_length(self, part_id):
        b, e = int(self.part_begin[part_id]), int(self.part_end[part_id])
        if e <= b:
            return 0
        if self.lengths_host is None:
            return self.max_length
        return int(self.lengths_host[e - 1])


class Searcher:
    """CudaSW4 (cudasw4.cuh:496-839) for ONE GPU; multi-GPU = one Searcher per process + host merge."""

    def __init__(self, device=0, num_top=10, matrix=None, kernel_types=None, gop=-11, gex=-1,
                 max_temp_bytes=4 << 30, merge_partitions=True):
        self.device = torch.device("cuda", device)
        self.ctx = capi.Context(device)
        self.num_top = int(num_top)
        self.kernel_types = kernel_types or KernelTypeConfig()
        self.kernel_types.validate()
        self.gop, self.gex = int(gop), int(gex)
        self.max_temp_bytes = int(max_temp_bytes)
        self.merge_partitions = merge_partitions
        self.db = None
        if matrix is not None:
            self.set_matrix(matrix)
        self._temps = {}           # one border scratch per concurrently running launch
        self._side_streams = []
        self.concurrent_runs = True  # launch the small long-subject runs on side streams next to the bulk run
        self._topk_temp = None
        self.record_kernel_events = False  # bench.py: HIP events around every DP launch
        self.kernel_events = []

    # -- configuration -------------------------------------------------------------------------
    def set_matrix(self, m21):  # setBlosum, cudasw4.cuh:570-572 / blosum.cu:21-119
        self.ctx.set_matrix(m21)

    def set_num_top(self, k):  # cudasw4.cuh:574-587
        self.num_top = int(k)

    def set_database(self, db):  # cudasw4.cuh:552-568
        if db.device != self.device:
            raise ValueError("DB is on %s, searcher on %s" % (db.device, self.device))
        self.db = db
        n = db.num_sequences
        self.scores = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
        self.ids = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        self.ovf_pos = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        self.ovf_count = torch.zeros(NUM_PARTITIONS + 4, dtype=torch.int32, device=self.device)  # one counter per packed run
        self._plan = self._launch_plan()

    def _launch_plan(self):
        """Partition walk of runAlignmentKernels (cudasw4.cuh:1742-2103): largest partition first.
        Adjacent partitions that use the same arithmetic kind are merged into one launch: the kernels
        take any subject length, so the partition only decides the kind.  The merged walk is the C++ driver's
        own planner (plan_launch_runs through swdrv_plan_runs) — one implementation, no drift; merge_partitions=False
        keeps one launch per non-empty partition like the reference (tests)."""
        db, kt = self.db, self.kernel_types
        if self.merge_partitions and db.lengths_host is not None:
            from . import driver
            return driver.plan_runs(db.lengths_host, kt.single_pass, kt.many_pass_small, kt.many_pass_large)
        runs = []
        for pid in range(NUM_PARTITIONS - 1, -1, -1):
            b, e = int(db.part_begin[pid]), int(db.part_end[pid])
            if e > b:
                runs.append({"kind": kt.kind_for_partition(pid), "part_id": pid, "begin": b, "end": e,
                             "maxlen": db.partition_max_length(pid)})
        return runs

    def _ensure_temp(self, nbytes, slot=0):
        nbytes = min(int(nbytes), self.max_temp_bytes)
        if nbytes <= 0:
            return 0, 0
        t = self._temps.get(slot)
        if t is None or t.numel() < nbytes:
            t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._temps[slot] = t
        return t.data_ptr(), t.numel()

    # -- the scan ------------------------------------------------------------------------------
    def scan(self, query_codes, timed=True, sync=True):
        """CudaSW4::scan (cudasw4.cuh:698-765): all subjects of the resident DB against one query.

        sync=False enqueues the scan and returns without waiting (bench.py's timed region brackets
        many scans with one synchronize); results are then read with `finish(result)`."""
        db = self.db
        if db is None:
            raise RuntimeError("set_database first")
        q = np.ascontiguousarray(query_codes, dtype=np.int8)
        stream = torch.cuda.current_stream(self.device)
        sp = stream.cuda_stream
        timed = timed and sync
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
        self.ctx.set_query(q, sp)
        self.ovf_count.zero_()
        self.scores.fill_(-1.0)  # cudasw4.cuh:405-409
        packed_runs = []  # (plan index, overflow list slot): every packed run has its own list and re-score launch
        # The reference round-robins partition launches over 10 work streams (cudasw4.cuh:293,1745-1748) so
        # that small partitions overlap.  Here: the run with the most subjects stays on the caller's stream,
        # the (few, long-subject) others go to side streams and are launched FIRST so that they hold their
        # handful of workgroups while the bulk run fills the rest of the GPU.
        plan = self._plan
        main_idx = max(range(len(plan)), key=lambda i: plan[i]["end"] - plan[i]["begin"]) if plan else 0
        use_side = self.concurrent_runs and len(plan) > 1
        order = [i for i in range(len(plan)) if i != main_idx] + ([main_idx] if plan else [])
        if use_side:
            while len(self._side_streams) < len(plan) - 1:
                self._side_streams.append(torch.cuda.Stream(device=self.device))
            fork = torch.cuda.Event()
            fork.record(stream)
        joins = []
        for slot, i in enumerate(order):
            run = plan[i]
            kind = run["kind"]
            n = run["end"] - run["begin"]
            on_side = use_side and i != main_idx
            st = self._side_streams[slot] if on_side else stream
            if on_side:
                st.wait_event(fork)
            tptr, tbytes = self._ensure_temp(self.ctx.scan_temp_bytes(kind, run["part_id"], n, run["maxlen"]),
                                             slot + 1 if on_side else 0)
            ovf_check = 1 if kind in capi.MAX_ACC else 0
            lst = len(packed_runs)
            if ovf_check:
                packed_runs.append((i, lst))
            if self.record_kernel_events:
                k0 = torch.cuda.Event(enable_timing=True)
                k0.record(st)
            self.ctx.scan_partition(kind, run["part_id"], db.chars.data_ptr(), db.offsets.data_ptr(),
                                    db.lengths.data_ptr(), run["begin"], n, run["maxlen"], self.gop, self.gex,
                                    self.scores.data_ptr(), self.ids.data_ptr(), db.id_offset,
                                    self.ovf_pos.data_ptr() + 4 * run["begin"], self.ovf_count.data_ptr() + 4 * lst,
                                    ovf_check, tptr, tbytes, st.cuda_stream)
            if self.record_kernel_events:
                k1 = torch.cuda.Event(enable_timing=True)
                k1.record(st)
                cells = float(len(q)) * (float(db.total_residues) if len(plan) == 1 else
                                         float(db.lengths_host[run["begin"]:run["end"]].sum()))
                self.kernel_events.append((k0, k1, cells))
            if on_side:
                j = torch.cuda.Event()
                j.record(st)
                joins.append(j)
        for j in joins:
            stream.wait_event(j)
        for i, lst in packed_runs:  # cudasw4.cuh:2117-2172; the group shape follows the run's longest subject
            run = plan[i]
            okind = self.kernel_types.overflow
            n = run["end"] - run["begin"]
            tptr, tbytes = self._ensure_temp(self.ctx.scan_temp_bytes(okind, -1, n, run["maxlen"]))
            self.ctx.rescore_overflow(okind, self.ovf_pos.data_ptr() + 4 * run["begin"], self.ovf_count.data_ptr() + 4 * lst, n,
                                      db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(),
                                      run["maxlen"], self.gop, self.gex, self.scores.data_ptr(), self.ids.data_ptr(),
                                      db.id_offset, tptr, tbytes, sp)
        k = min(self.num_top, db.num_sequences)
        top_scores = top_ids = None
        if k > 0:  # cudasw4.cuh:1357-1401
            need = capi.topk_temp_bytes(db.num_sequences, k)
            if self._topk_temp is None or self._topk_temp.numel() < need:
                self._topk_temp = torch.empty(need, dtype=torch.uint8, device=self.device)
            out_s = torch.empty(k, dtype=torch.float32, device=self.device)
            out_i = torch.empty(k, dtype=torch.int32, device=self.device)
            self.ctx.topk(self.scores.data_ptr(), self.ids.data_ptr(), db.num_sequences, k, out_s.data_ptr(),
                          out_i.data_ptr(), self._topk_temp.data_ptr(), self._topk_temp.numel(), sp)
            top_scores, top_ids = out_s, out_i
        cells = float(len(q)) * float(db.total_residues)
        res = ScanResult(scores=np.zeros(0, np.int32), reference_ids=np.zeros(0, np.int64), num_overflows=-1)
        # the counters are reused by the next scan: keep this scan's values (a stream-ordered copy)
        res.stats = {"cells": cells, "launches": len(self._plan), "_top": (top_scores, top_ids), "_ovf": self.ovf_count.clone()}
        if not sync:
            return res
        if timed:
            ev1.record(stream)
            ev1.synchronize()
            res.seconds = ev0.elapsed_time(ev1) * 1e-3
            res.gcups = cells / 1e9 / res.seconds if res.seconds > 0 else 0.0  # cudasw4.cuh:2264-2271
        else:
            stream.synchronize()
        return self.finish(res)

    def finish(self, res):
        """Copy the top-K and the overflow count of an enqueued scan to the host (D2H of cudasw4.cuh:1465-1487)."""
        top_scores, top_ids = res.stats.pop("_top", (None, None))
        if top_scores is not None:
            res.scores = top_scores.cpu().numpy().astype(np.int32)
            res.reference_ids = top_ids.cpu().numpy().astype(np.int64)
        ovf = res.stats.pop("_ovf", None)
        res.num_overflows = int(ovf.sum().item()) if ovf is not None else -1
        return res

    def all_scores(self):
        """Every subject's score of the last scan (the CUDASW_DEBUG_CHECK_CORRECTNESS view,
        cudasw4.cuh:728-756)."""
        return self.scores[:self.db.num_sequences].cpu().numpy().astype(np.int32)


def shard_ranges(offsets, lengths, world):
    """partitionDBAmongstGpus (cudasw4.cuh:928-1004): every length partition of the (length-sorted) DB is cut into
    <= world contiguous, char-balanced subject ranges, so each rank sees every length class.  Returns
    ranges[rank][partition] = (begin, end).  Computed by the C++ driver's shard_database (csrc/host/db_format.cpp)."""
    from . import driver
    return driver.shard_ranges(offsets, lengths, world)


def build_shard(chars, offsets, lengths, ranges):
    """Concatenate the per-partition ranges of one rank into a dbdata-layout shard (still sorted by
    length).  Returns (chars, offsets, lengths, global_ids)."""
    chars = np.asarray(chars, dtype=np.int8)
    offsets = np.asarray(offsets, dtype=np.uint64)
    lengths = np.asarray(lengths, dtype=np.int32)
    c_parts, l_parts, ids = [], [], []
    for (b, e) in ranges:
        if e <= b:
            continue
        c_parts.append(chars[int(offsets[b] - offsets[0]):int(offsets[e] - offsets[0])])
        l_parts.append(lengths[b:e])
        ids.append(np.arange(b, e, dtype=np.int64))
    if not l_parts:
        return np.zeros(0, np.int8), np.zeros(1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.int64)
    sl = np.concatenate(l_parts)
    so = np.zeros(len(sl) + 1, dtype=np.uint64)
    so[1:] = np.cumsum((sl.astype(np.int64) + 3) // 4 * 4)
    return np.concatenate(c_parts), so, sl, np.concatenate(ids)


def merge_topk(per_rank, k):
    """Host-side merge of per-GPU top-K lists (replaces the peer-copy gather + sort on device 0,
    cudasw4.cuh:1415-1463).  per_rank: list of (scores, global_ids); score desc, id asc on ties."""
    if not per_rank:
        return np.zeros(0, np.int32), np.zeros(0, np.int64)
    s = np.concatenate([np.asarray(a, dtype=np.int64) for a, _ in per_rank])
    i = np.concatenate([np.asarray(b, dtype=np.int64) for _, b in per_rank])
    order = np.lexsort((i, -s))[:k]
    return s[order].astype(np.int32), i[order]
