"""ctypes binding of the C ABI in include/cudasw4_amd.h (libcudasw4_amd.so).

This is plumbing: every compute call goes to the HIP library.  There is no Python or CPU fallback —
if the shared library is missing the import of this module raises, and without a GPU
`Context()` raises `SwError` (SW_ERR_NO_DEVICE).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CUDASW4_AMD_LIB", os.path.join(_HERE, "lib", "libcudasw4_amd.so"))  # env override: A/B builds

KIND_F16X2, KIND_I16X2, KIND_I32, KIND_F32 = 0, 1, 2, 3
KIND_NAMES = {"half2": KIND_F16X2, "f16x2": KIND_F16X2, "dpxs16": KIND_I16X2, "i16x2": KIND_I16X2,
              "dpxs32": KIND_I32, "i32": KIND_I32, "float": KIND_F32, "f32": KIND_F32}
MAX_ACC = {KIND_F16X2: 2048, KIND_I16X2: 25000}

SW_OK = 0
ERR_NAMES = {-1: "SW_ERR_INVALID", -2: "SW_ERR_HIP", -3: "SW_ERR_NO_QUERY", -4: "SW_ERR_NO_MATRIX",
             -5: "SW_ERR_TEMP", -6: "SW_ERR_NO_DEVICE"}

# every symbol include/cudasw4_amd.h (the boundary) and include/cudasw4_amd_engine.h (the batch engine's building blocks) declare
EXPORTS = ["sw_batch_create", "sw_batch_destroy", "sw_scan_batch", "sw_batch_join", "sw_batch_side_events", "sw_batch_feedback",
           "sw_batch_handshake_active", "sw_batch_stats", "sw_batch_signal_state", "sw_batch_open_gates", "sw_batch_reset",
           "sw_batch_test_lose_side_launch", "sw_query_length", "sw_batch_describe_plan",
           "sw_version", "sw_last_error", "sw_device_count", "sw_ctx_create", "sw_ctx_destroy", "sw_set_matrix",
           "sw_set_query", "sw_scan_temp_bytes", "sw_scan_partition", "sw_rescore_overflow", "sw_rescore_overflow_stat",
           "sw_topk_temp_bytes",
           "sw_topk", "sw_plan_query", "sw_check_letter_codes", "sw_plan_launch", "sw_set_start_signal",
           "sw_window_overlap", "sw_reduce_windows", "sw_rescore_service", "sw_rescore_overflow_claim",
           "sw_rescore_service_temp_bytes", "sw_streams_run_concurrently", "sw_set_dry_signal", "sw_set_dirty_counter", "sw_set_grid_reserve",
           "sw_set_long16_min", "sw_scan_rows_pipelined",
           "sw_scan_rows_pipelined_temp_bytes", "sw_probe_handshake", "sw_launch_vgpr_slot",
           "sw_set_rows_pipeline_slot", "sw_rescore_overflow_pipelined", "sw_rescore_overflow_pipelined_temp_bytes", "sw_measure_valu_rate"]


class SwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "SW_ERR"), code, msg))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcudasw4_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C cudasw4_amd/csrc`): %s" % LIB_PATH)
    # PyTorch wheels bundle their own HIP runtime.  If this library pulled in the system's libamdhip64 first, a later
    # `import torch` would bring a second runtime into the process and find no GPU: load torch's first when it is there.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
    L.sw_version.restype = ctypes.c_char_p
    L.sw_last_error.restype = ctypes.c_char_p
    L.sw_device_count.restype = ctypes.c_int
    L.sw_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.sw_ctx_destroy.argtypes = [vp]
    L.sw_set_matrix.argtypes = [vp, vp, ctypes.c_int]
    L.sw_set_query.argtypes = [vp, vp, i32, vp]
    L.sw_scan_temp_bytes.restype = sz
    L.sw_scan_temp_bytes.argtypes = [vp, ctypes.c_int, ctypes.c_int, i32, i32]
    L.sw_scan_partition.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, i32, i32, i32, ctypes.c_int,
                                    ctypes.c_int, vp, vp, i64, vp, vp, ctypes.c_int, vp, sz, vp]
    L.sw_rescore_overflow.argtypes = [vp, ctypes.c_int, vp, vp, i32, vp, vp, vp, i32, ctypes.c_int, ctypes.c_int,
                                      vp, vp, i64, vp, sz, vp]
    L.sw_rescore_overflow_stat.argtypes = [vp, ctypes.c_int, vp, vp, i32, vp, vp, vp, i32, ctypes.c_int, ctypes.c_int,
                                           vp, vp, i64, vp, sz, i32, vp, vp]
    L.sw_topk_temp_bytes.restype = sz
    L.sw_topk_temp_bytes.argtypes = [i64, ctypes.c_int]
    L.sw_topk.argtypes = [vp, vp, vp, i64, ctypes.c_int, vp, vp, vp, sz, vp]
    L.sw_plan_query.argtypes = [ctypes.c_int, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.sw_check_letter_codes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    L.sw_set_start_signal.argtypes = [vp, vp]
    L.sw_probe_handshake.argtypes = [vp, vp, vp, vp]
    L.sw_rescore_overflow_claim.argtypes = [vp, ctypes.c_int, vp, vp, i32, vp, vp, vp, i32, ctypes.c_int, ctypes.c_int, vp, vp, i64, vp, sz, i32, vp, vp]
    L.sw_rescore_overflow_pipelined_temp_bytes.restype = sz
    L.sw_rescore_overflow_pipelined_temp_bytes.argtypes = [vp, i32]
    L.sw_rescore_overflow_pipelined.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, i32, ctypes.c_int, ctypes.c_int, vp, vp, i64, vp, i32, vp, vp, sz, vp]
    L.sw_measure_valu_rate.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.sw_launch_vgpr_slot.argtypes = [vp, ctypes.c_int, ctypes.c_int, i32, i32]
    L.sw_set_rows_pipeline_slot.argtypes = [vp, ctypes.c_int]
    L.sw_set_dry_signal.argtypes = [vp, vp, ctypes.c_uint32]
    L.sw_set_grid_reserve.argtypes = [vp, i32]
    L.sw_set_long16_min.argtypes = [vp, i32]
    L.sw_scan_rows_pipelined_temp_bytes.restype = sz
    L.sw_scan_rows_pipelined_temp_bytes.argtypes = [vp, i32, i32]
    L.sw_scan_rows_pipelined.argtypes = [vp, vp, vp, vp, i32, i32, i32, ctypes.c_int, ctypes.c_int, vp, vp, i64, vp, vp, vp, i32, vp, sz, vp]
    L.sw_window_overlap.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.sw_window_overlap.restype = i32
    L.sw_reduce_windows.argtypes = [vp, vp, vp, vp, i32, vp, vp, ctypes.c_int64, vp]
    L.sw_plan_launch.argtypes = [vp, ctypes.c_int, ctypes.c_int, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32),
                                 ctypes.POINTER(i32), ctypes.POINTER(i32)]
    return L


lib = _load()


def check(rc):
    if rc != SW_OK:
        raise SwError(rc, lib.sw_last_error().decode())


def version():
    return lib.sw_version().decode()


def device_count():
    return int(lib.sw_device_count())


def plan_query(kind, qlen):
    r, s = ctypes.c_int32(), ctypes.c_int32()
    check(lib.sw_plan_query(kind, qlen, ctypes.byref(r), ctypes.byref(s)))
    return r.value, s.value


class Context:
    """One per GPU (sw_ctx).  Pointers are passed as integers (tensor.data_ptr())."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        check(lib.sw_ctx_create(device, ctypes.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if self.handle:
            lib.sw_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_matrix(self, m21):
        import numpy as np
        m = np.ascontiguousarray(m21, dtype=np.int8).reshape(-1)
        dim = int(round(len(m) ** 0.5))
        check(lib.sw_set_matrix(self.handle, m.ctypes.data, dim))

    def set_query(self, codes, stream=0):
        import numpy as np
        q = np.ascontiguousarray(codes, dtype=np.int8)
        check(lib.sw_set_query(self.handle, q.ctypes.data, len(q), stream))

    def scan_temp_bytes(self, kind, part_id, n, max_subject_len):
        return int(lib.sw_scan_temp_bytes(self.handle, kind, part_id, n, max_subject_len))

    def scan_partition(self, kind, part_id, chars, offsets, lengths, first_pos, n, max_subject_len, gop, gex,
                       scores, ids, id_offset=0, ovf_pos=0, ovf_count=0, ovf_check=0, temp=0, temp_bytes=0, stream=0):
        check(lib.sw_scan_partition(self.handle, kind, part_id, chars, offsets, lengths, first_pos, n, max_subject_len,
                                    gop, gex, scores, ids, id_offset, ovf_pos, ovf_count, ovf_check, temp, temp_bytes,
                                    stream))

    def measure_valu_rate(self, mix, millis=50):
        """-> (lane-instructions per second the device issues of that instruction mix, shader clock in Hz seen by the waves)"""
        r, hz = ctypes.c_double(), ctypes.c_double()
        check(lib.sw_measure_valu_rate(self.handle, mix, millis, ctypes.byref(r), ctypes.byref(hz)))
        return r.value, hz.value

    def launch_vgpr_slot(self, kind, part_id, n, max_subject_len):
        return int(lib.sw_launch_vgpr_slot(self.handle, kind, part_id, n, max_subject_len))

    def set_rows_pipeline_slot(self, vgprs):
        check(lib.sw_set_rows_pipeline_slot(self.handle, vgprs))

    def scan_rows_pipelined_temp_bytes(self, n, max_subject_len):
        return int(lib.sw_scan_rows_pipelined_temp_bytes(self.handle, n, max_subject_len))

    def scan_rows_pipelined(self, chars, offsets, lengths, first_pos, n, max_subject_len, gop, gex, scores, ids, id_offset=0,
                            fail_count=0, temp=0, temp_bytes=0, stream=0, over_limit_count=0, over_limit_count2=0, packed_limit=0):
        """Very long subjects as pipelines of one-wave stages across many CUs (sw_scan_rows_pipelined)."""
        check(lib.sw_scan_rows_pipelined(self.handle, chars, offsets, lengths, first_pos, n, max_subject_len, gop, gex, scores,
                                         ids, id_offset, fail_count, over_limit_count, over_limit_count2, packed_limit, temp,
                                         temp_bytes, stream))

    def rescore_overflow_pipelined_temp_bytes(self, max_subject_len):
        return int(lib.sw_rescore_overflow_pipelined_temp_bytes(self.handle, max_subject_len))

    def rescore_overflow_pipelined(self, ovf_pos, ovf_count, max_count, chars, offsets, lengths, max_subject_len, min_subject_len,
                                   gop, gex, scores, ids, id_offset, fail_count, packed_limit, true_count, temp, temp_bytes, stream=0):
        check(lib.sw_rescore_overflow_pipelined(self.handle, ovf_pos, ovf_count, max_count, chars, offsets, lengths, max_subject_len,
                                                min_subject_len, gop, gex, scores, ids, id_offset, fail_count, packed_limit,
                                                true_count, temp, temp_bytes, stream))

    def rescore_overflow_claim(self, kind, ovf_pos, ovf_count, max_count, chars, offsets, lengths, max_subject_len, gop, gex,
                               scores, ids, id_offset, temp, temp_bytes, packed_limit, true_count, stream=0):
        check(lib.sw_rescore_overflow_claim(self.handle, kind, ovf_pos, ovf_count, max_count, chars, offsets, lengths,
                                            max_subject_len, gop, gex, scores, ids, id_offset, temp, temp_bytes, packed_limit,
                                            true_count, stream))

    def rescore_overflow(self, kind, ovf_pos, ovf_count, max_count, chars, offsets, lengths, max_subject_len, gop, gex,
                         scores, ids, id_offset=0, temp=0, temp_bytes=0, stream=0):
        check(lib.sw_rescore_overflow(self.handle, kind, ovf_pos, ovf_count, max_count, chars, offsets, lengths,
                                      max_subject_len, gop, gex, scores, ids, id_offset, temp, temp_bytes, stream))

    def rescore_overflow_stat(self, kind, ovf_pos, ovf_count, max_count, chars, offsets, lengths, max_subject_len, gop, gex,
                              scores, ids, id_offset, temp, temp_bytes, packed_limit, true_count, stream=0):
        check(lib.sw_rescore_overflow_stat(self.handle, kind, ovf_pos, ovf_count, max_count, chars, offsets, lengths,
                                           max_subject_len, gop, gex, scores, ids, id_offset, temp, temp_bytes,
                                           packed_limit, true_count, stream))

    def plan_launch(self, kind, part_id, n, max_subject_len):
        """-> (kind computed in, rows per lane, query stripes, lanes per group) of the launch sw_scan_partition
        (part_id >= 0) or sw_rescore_overflow (part_id = -1) would make for the current query."""
        k, r, s, l = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        check(lib.sw_plan_launch(self.handle, kind, part_id, n, max_subject_len, ctypes.byref(k), ctypes.byref(r),
                                 ctypes.byref(s), ctypes.byref(l)))
        return k.value, r.value, s.value, l.value

    def check_letter_codes(self, chars, n, bad_flag, stream=0):
        check(lib.sw_check_letter_codes(self.handle, chars, n, bad_flag, stream))

    def topk(self, scores, ids, n, k, out_scores, out_ids, temp, temp_bytes, stream=0):
        check(lib.sw_topk(self.handle, scores, ids, n, k, out_scores, out_ids, temp, temp_bytes, stream))


def topk_temp_bytes(n, k):
    return int(lib.sw_topk_temp_bytes(n, k))
