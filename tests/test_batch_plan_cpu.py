"""CPU: what sw_scan_batch PLANS for a batch (cudasw4_amd/csrc/sw_batch.hip: BatchJob::plan, through sw_batch_describe_plan),
on the fake runtime of tests/host/fake_gpu, which compiles the real engine: which subjects leave the scan launches for the
pipelines on a whole DB and on a 1/8 shard, what becomes of a small partition 34, and the switch that turns pipelines off.
Host arithmetic only — nothing is launched."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

ROOT = O.ROOT
FAKE_DIR = os.path.join(ROOT, "tests", "host", "fake_gpu")
FAKE_LIB = os.path.join(ROOT, "tests", "host", "_build", "libfake_driver.so")

SCRIPT = r'''
import ctypes, json, os, sys
import numpy as np
L = ctypes.CDLL(%(lib)r)
vp, i32p, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64)

class Args(ctypes.Structure):
    _fields_ = [("kinds", ctypes.c_int * 4), ("chars", vp), ("offsets", vp), ("lengths", vp), ("n", ctypes.c_int32),
                ("part_begin", i32p), ("part_maxlen", i32p), ("long_lengths", i32p), ("long_offsets", u64p),
                ("long_offsets_bias", ctypes.c_uint64), ("batch_bytes", ctypes.c_uint64), ("gop", ctypes.c_int), ("gex", ctypes.c_int),
                ("scores", vp), ("ids", vp), ("id_offset", ctypes.c_int64), ("ovf_pos", vp), ("counters", vp), ("zero_counters", ctypes.c_int),
                ("max_temp_bytes", ctypes.c_size_t), ("stream", vp), ("work_slot", ctypes.c_int), ("allow_service", ctypes.c_int),
                ("arm_signal", vp), ("arm_value", ctypes.c_uint32), ("wait_signal", vp), ("wait_value", ctypes.c_uint32),
                ("grid_reserve_side", ctypes.c_int32), ("alt_side_stream", ctypes.c_int), ("records", vp), ("records_cap", ctypes.c_int32),
                ("records_used", vp), ("record_mode", ctypes.c_int)]

L.sw_last_error.restype = ctypes.c_char_p
ctx, eng = vp(), vp()
assert L.sw_ctx_create(0, ctypes.byref(ctx)) == 0
assert L.sw_batch_create(ctx, None, ctypes.byref(eng)) == 0

def plan(lengths, qlen, kinds=(1, 1, 2, 2)):
    lengths = np.sort(np.asarray(lengths, dtype=np.int32))
    bounds = [64 * (i + 1) for i in range(16)] + [1024 + 0 * i for i in range(0)]
    # the reference's 36 partitions end at 64, 128, ... , 1280 (every 64 up to 1024, then 1088 .. 1280?) — this test only needs
    # partitions 0..33 = everything up to 1280, 34 = 1281..8000, 35 = above: positions of the two long partitions
    b34, b35 = int(np.searchsorted(lengths, 1280, side="right")), int(np.searchsorted(lengths, 8000, side="right"))
    n = len(lengths)
    pb = np.zeros(37, dtype=np.int32)
    pb[1:34] = np.linspace(0, b34, 34).astype(np.int32)[1:]
    pb[34], pb[35], pb[36] = b34, b35, n
    pmax = np.zeros(36, dtype=np.int32)
    for p in range(36):
        if pb[p + 1] > pb[p]:
            pmax[p] = lengths[pb[p + 1] - 1]
    longl = np.ascontiguousarray(lengths[b34:])
    longo = np.zeros(len(longl), dtype=np.uint64)
    q = np.zeros(qlen, dtype=np.int8)
    assert L.sw_set_query(ctx, q.ctypes.data_as(vp), qlen, None) == 0
    a = Args()
    a.kinds[:] = kinds
    a.n = n
    a.part_begin = pb.ctypes.data_as(i32p); a.part_maxlen = pmax.ctypes.data_as(i32p)
    a.long_lengths = longl.ctypes.data_as(i32p); a.long_offsets = longo.ctypes.data_as(u64p)
    a.batch_bytes = int(lengths.astype(np.int64).sum())
    a.gop, a.gex = -11, -1
    a.allow_service = 1
    buf = ctypes.create_string_buffer(4096)
    rc = L.sw_batch_describe_plan(eng, ctypes.byref(a), buf, 4096)
    assert rc == 0, L.sw_last_error()
    return {"text": buf.value.decode(), "b34": b34, "b35": b35, "n": n}

rng = np.random.default_rng(1)
def db(n, n34, giants):
    return np.concatenate([rng.integers(30, 1281, n), rng.integers(1281, 6000, n34), np.array(giants, dtype=np.int64)])
out = {"whole": plan(db(560000, 9000, [9000, 12000, 35213]), 5478),
       "shard": plan(db(70000, 1500, [35213]), 5478),
       "shard_short_query": plan(db(70000, 1500, [35213]), 144),
       "small34": plan(db(70000, 200, [35213]), 2005)}
# argument errors (checked before anything is planned or launched)
def err(mut, query=True):
    lengths = np.sort(db(2000, 50, [9000])).astype(np.int32)
    b34, b35 = int(np.searchsorted(lengths, 1280, side="right")), int(np.searchsorted(lengths, 8000, side="right"))
    pb = np.zeros(37, dtype=np.int32); pb[1:34] = np.linspace(0, b34, 34).astype(np.int32)[1:]; pb[34], pb[35], pb[36] = b34, b35, len(lengths)
    pmax = np.zeros(36, dtype=np.int32)
    for p in range(36):
        if pb[p + 1] > pb[p]:
            pmax[p] = lengths[pb[p + 1] - 1]
    a = Args(); a.kinds[:] = (1, 1, 2, 2); a.n = len(lengths)
    a.part_begin = pb.ctypes.data_as(i32p); a.part_maxlen = pmax.ctypes.data_as(i32p)
    a.batch_bytes = int(lengths.sum()); a.gop, a.gex = -11, -1
    keep = mut(a, pmax)
    buf = ctypes.create_string_buffer(1024)
    rc = L.sw_batch_describe_plan(eng, ctypes.byref(a), buf, 1024)
    return [rc, L.sw_last_error().decode() if rc else ""]
def bad_kind(a, pmax): a.kinds[0] = 7
def unpacked_small(a, pmax): a.kinds[1] = 3
def packed_overflow(a, pmax): a.kinds[3] = 0
def nominal_boundary(a, pmax): pmax[35] = 2**31 - 1
def no_tables(a, pmax): a.part_begin = None
def fine(a, pmax): pass
out["errors"] = {"bad_kind": err(bad_kind), "unpacked_small": err(unpacked_small), "packed_overflow": err(packed_overflow),
                 "nominal_boundary": err(nominal_boundary), "no_tables": err(no_tables), "fine": err(fine)}
ctx2, eng2 = vp(), vp()
assert L.sw_ctx_create(0, ctypes.byref(ctx2)) == 0 and L.sw_batch_create(ctx2, None, ctypes.byref(eng2)) == 0
eng_saved, eng = eng, eng2          # an engine whose context has no query yet
out["errors"]["no_query"] = err(fine)
eng = eng_saved
print("RESULT " + json.dumps(out))
'''


@pytest.fixture(scope="module")
def fake_lib():
    p = subprocess.run(["make", "-C", FAKE_DIR], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return FAKE_LIB


def run(fake_lib, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, "-c", SCRIPT % {"lib": fake_lib}], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def parts(text):
    return [t.strip() for t in text.split(";") if t.strip()]


def test_what_a_whole_db_and_a_shard_of_it_launch(fake_lib):
    r = run(fake_lib)
    whole, shard = r["whole"], r["shard"]
    # a whole DB: 570 000 subjects keep the bulk launch busy for ~0.2 s — longer than a wave-wide group needs for the
    # 35 213-residue giant: nothing is pipelined, partition 34 (9 000 subjects >= the merge minimum) joins the bulk launch,
    # partition 35 is a side launch of the 32-bit kind
    w = parts(whole["text"])
    assert not [t for t in w if t.startswith("pipeline")]
    assert "bulk kind 1 p33 [0,%d)" % whole["b35"] in whole["text"]
    assert "side kind 2 p35 [%d,%d) maxlen 35213" % (whole["b35"], whole["n"]) in whole["text"]
    # 1/8 of it with the same giant: the bulk launch is over in 15 ms, the longest subjects of partition 34 (at most 256)
    # and all of partition 35 leave it
    s = parts(shard["text"])
    p34 = [t for t in s if t.startswith("pipeline p34")]
    assert len(p34) == 1
    begin = int(p34[0].split("[")[1].split(",")[0])
    assert shard["b35"] - 256 <= begin < shard["b35"] and p34[0].split(")")[0].endswith(",%d" % shard["b35"])
    assert "pipeline p35 [%d,%d)" % (shard["b35"], shard["n"]) in shard["text"]
    assert "bulk kind 1 p33 [0,%d)" % begin in shard["text"]          # what is left of partition 34 (>= 512 subjects) merges into the bulk
    # a 144-residue query on the same shard pipelines as many as the cap allows
    q = parts(r["shard_short_query"]["text"])
    assert [t for t in q if t.startswith("pipeline p34")][0].startswith("pipeline p34 [%d," % (shard["b35"] - 256))
    # a partition 34 below the merge minimum keeps a side launch of its own
    m = parts(r["small34"]["text"])
    side = [t for t in m if t.startswith("side kind 1 p34")]
    assert len(side) == 1 and side[0].startswith("side kind 1 p34 [%d," % r["small34"]["b34"])
    assert [t for t in m if t.startswith("bulk kind 1") and "[0,%d)" % r["small34"]["b34"] in t]


def test_pipelines_switched_off(fake_lib):
    r = run(fake_lib, {"CUDASW4_AMD_PIPELINES": "0"})
    for k in ("whole", "shard", "shard_short_query", "small34"):
        assert "pipeline" not in r[k]["text"], (k, r[k]["text"])
    assert "side kind 2 p35" in r["shard"]["text"]      # the giants: an ordinary side launch of the 32-bit kind


def test_argument_errors_are_named(fake_lib):
    e = run(fake_lib)["errors"]
    assert e["fine"][0] == 0
    for k in ("bad_kind", "unpacked_small", "packed_overflow", "nominal_boundary", "no_tables", "no_query"):
        assert e[k][0] != 0, k
    assert "unknown kind" in e["bad_kind"][1]
    assert "manyPass_small must be a packed kind" in e["unpacked_small"][1] and "manyPass_small" in e["packed_overflow"][1]
    assert "longest subject of the partition" in e["nominal_boundary"][1] and "2147483647" in e["nominal_boundary"][1]
    assert "partition tables missing" in e["no_tables"][1]
    assert "sw_set_query" in e["no_query"][1]
