"""CPU: the REAL C++ host driver (cudasw4_amd/csrc/host/search_driver.cpp) on TWO DISTINCT device ordinals.

No multi-GPU box is available to the builder, and every GPU test drives several shards of device 0.  What has never run is
the code that distinguishes devices: per-device hipSetDevice in the worker threads, streams / events / buffers created on
one device and used on it only, the shard-to-device mapping, the per-device start handshake, and the host merge of lists
that come from different devices.  tests/host/fake_gpu builds the driver against a host-memory fake of the HIP runtime
with two devices that COUNTS every use of an object while another device is current, and a fake of the C ABI whose
"scores" are a deterministic function of subject and query (test infrastructure: nothing of it is shipped or measured).
"""
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
sys.path.insert(0, %(root)r)
from cudasw4_amd import driver, synthdb
L = driver.lib
for f in ("fake_hip_violations", "fake_sw_bad_owner"):
    getattr(L, f).restype = ctypes.c_int
L.fake_hip_violation_text.restype = ctypes.c_char_p
L.fake_hip_stream_calls.restype = ctypes.c_long
L.fake_hip_stream_calls.argtypes = [ctypes.c_int]
L.fake_sw_scans.restype = ctypes.c_long
L.fake_sw_scans.argtypes = [ctypes.c_int]
L.fake_sw_rescored.restype = ctypes.c_long
L.fake_sw_rescored.argtypes = [ctypes.c_int]
L.fake_sw_dry_signals.restype = ctypes.c_long
L.fake_sw_dry_signals.argtypes = [ctypes.c_int]

lengths = synthdb.sprot_like_lengths(30000, seed=3, max_len=12000)
chars, offsets, lengths = synthdb.random_db(lengths, seed=4)
queries = [b"MKTAYIAKQRQISFVKSHFSRQ", b"ACDEFGHIKLMNPQRSTVWY" * 30, b"W" * 700]
out = {}
for name, devices, kw in (("one", [0], {}), ("two", [0, 1], {}), ("rev", [1, 0], {}),
                          ("two_streamed", [0, 1], dict(max_gpu_mem=1, max_batch_bytes=1 << 20)),
                          ("two_hybrid", [1, 0], dict(max_gpu_mem=(4 << 20) + 24 * 15000, max_batch_bytes=1 << 19))):
    d = driver.Driver(devices=devices, num_top=25, kinds=(1, 1, 2, 2), **kw)
    d.db_from_arrays(chars, offsets, lengths)
    d.upload()
    res = []
    for q in queries:
        r = d.scan(q)
        ids, sc = d.all_scores()
        order = np.argsort(ids)
        res.append({"scores": r["scores"].tolist(), "ids": r["ids"].tolist(), "rescored": r["num_rescored"],
                    "all": sc[order].tolist(), "ids_cover": bool((ids[order] == np.arange(len(lengths))).all())})
    pipelined = d.scan_many(queries)
    ids, sc = d.all_scores()   # of the query collected last, whichever lane it ran on
    overlaps = d.tail_overlaps()
    by_rule = d.scan_stream(queries)   # align's loop: two in flight where the driver's rule says so
    out[name] = {"res": res, "pipelined": [[p["scores"].tolist(), p["ids"].tolist()] for p in pipelined],
                 "last_all_after_pipelined": sc[np.argsort(ids)].tolist(), "tail_overlaps": overlaps,
                 "by_rule": [[p["scores"].tolist(), p["ids"].tolist()] for p in by_rule],
                 "devices": [d.device_of(g) for g in range(d.num_gpus())],
                 "subjects": [d.shard_info(g)["subjects"] for g in range(d.num_gpus())],
                 "numa": [d.numa_node(g) for g in range(d.num_gpus())]}
    d.close()
out["violations"] = L.fake_hip_violations()
out["violation_text"] = L.fake_hip_violation_text().decode()
out["bad_owner"] = L.fake_sw_bad_owner()
out["stream_calls"] = [L.fake_hip_stream_calls(0), L.fake_hip_stream_calls(1)]
out["scans"] = [L.fake_sw_scans(0), L.fake_sw_scans(1)]
out["rescored"] = [L.fake_sw_rescored(0), L.fake_sw_rescored(1)]
out["dry_signals"] = [L.fake_sw_dry_signals(0), L.fake_sw_dry_signals(1)]
print("RESULT " + json.dumps(out))
'''


@pytest.fixture(scope="module")
def fake_lib():
    p = subprocess.run(["make", "-C", FAKE_DIR], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return FAKE_LIB


def test_driver_on_two_distinct_devices(fake_lib):
    env = dict(os.environ, CUDASW4_AMD_HOST_LIB=fake_lib)
    p = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    # no object of one device was ever used while the other one was current, no buffer reached the wrong context, and
    # no stream waited for a start signal that had not been raised
    assert out["violations"] == 0, out["violation_text"]
    assert out["bad_owner"] == 0
    # both ordinals worked, in both orders
    assert out["two"]["devices"] == [0, 1] and out["rev"]["devices"] == [1, 0]
    assert all(c > 100 for c in out["stream_calls"]) and all(s > 10 for s in out["scans"])
    assert all(r > 0 for r in out["rescored"])            # the overflow / re-score plumbing ran on both devices
    one = out["one"]
    for name in ("two", "rev", "two_streamed", "two_hybrid"):
        run = out[name]
        assert len(run["subjects"]) == 2 and sum(run["subjects"]) == 30000 and min(run["subjects"]) > 12000
        for a, b in zip(one["res"], run["res"]):
            assert b["ids_cover"] and a["all"] == b["all"]                          # every score, by global id
            assert a["scores"] == b["scores"] and a["ids"] == b["ids"]              # the merged top-25
            assert a["rescored"] == b["rescored"] > 0
        assert run["pipelined"] == one["pipelined"] == [[r["scores"], r["ids"]] for r in one["res"]]
        assert run["last_all_after_pipelined"] == one["res"][-1]["all"]
        assert run["by_rule"] == one["pipelined"]
    # tail hand-over (two queries in flight on a resident shard): the second and third query of scan_many ran on the other
    # lane, gated on the dry signal of the query before — per GPU; never on a streamed or hybrid shard
    # (tail_overlaps was read before the scan_stream pass)
    assert one["tail_overlaps"] == 2 and out["two"]["tail_overlaps"] == 4 and out["rev"]["tail_overlaps"] == 4
    assert out["two_streamed"]["tail_overlaps"] == 0 and out["two_hybrid"]["tail_overlaps"] == 0
    assert all(n > 0 for n in out["dry_signals"])
    assert out["two"]["numa"] == [-1, -1]       # the fake's PCI ids exist nowhere: unknown node, no binding attempted


def test_numa_helpers():
    """cpus_of_numa_node / bind_thread_to_numa_node through the product library: node 0 exists on every Linux box; the
    binding keeps to the CPUs the process may use, and an unknown node changes nothing."""
    from cudasw4_amd import driver
    before = os.sched_getaffinity(0)
    try:
        assert driver.bind_to_numa_node(-1) is False and os.sched_getaffinity(0) == before
        assert driver.bind_to_numa_node(4096) is False and os.sched_getaffinity(0) == before
        if os.path.exists("/sys/devices/system/node/node0/cpulist"):
            assert driver.bind_to_numa_node(0) is True
            assert os.sched_getaffinity(0) <= before and len(os.sched_getaffinity(0)) >= 1
    finally:
        os.sched_setaffinity(0, before)
