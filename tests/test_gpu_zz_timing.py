"""GPU: everything that asserts on a WALL CLOCK or a measured rate, in the file that sorts behind every parity test.

The driver runs `pytest -m gpu -x`: one noisy box must not turn the oracle / golden comparisons behind a failed timing
assertion into "untested" (VERDICT r4 item 3).  So: the overlap of the shards' spans, the accounting behind bench.py's
roofline figures, the bounded wait of a pipeline stage whose neighbour was lost, and the host watchdog of a scan whose side
launch never starts."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_lib as O
from test_gpu_configs import FASTA, ROOT, run_bench, sprot_db  # noqa: F401  (sprot_db: module-scoped fixture)

pytestmark = pytest.mark.gpu


def test_streamed_shards_run_concurrently():
    """Config 5's route with three shards in flight (devices=[0,0,0], one worker thread per shard): every shard's span
    overlaps every other's — they start together, none waits for another to finish."""
    from cudasw4_amd import driver, synthdb
    lengths = synthdb.sprot_like_lengths(24000, seed=11, max_len=12000)
    chars, offsets, lengths = synthdb.random_db(lengths, seed=12, other_fraction=0.01)
    _, letters = O.read_fasta(FASTA)
    d = driver.Driver(devices=[0, 0, 0], num_top=25, kinds=(2, 1, 2, 2), max_gpu_mem=1, max_batch_bytes=400_000)
    d.db_from_arrays(chars, offsets, lengths)
    for qi in (7, 13, 19):
        d.scan(letters[qi])
        spans = d.gpu_spans()
        assert len(spans) == 3
        latest_begin, earliest_end = max(b for b, e in spans), min(e for b, e in spans)
        assert latest_begin < earliest_end, spans
    d.close()


def test_pipeline_lost_stage_gives_up_within_bounds():
    """sw_scan_rows_pipelined (csrc/sw_rows_pipeline.hpp): a stage that never produces (test hook) — its successors poll a
    bounded number of times, mark the subject -2 and count themselves in fail_count; the call returns in bounded time.
    Subjects whose pipelines were done before the abort keep their exact scores."""
    import torch
    from cudasw4_amd import capi, search
    from test_gpu_rows_pipeline import env, run_pipeline
    rng = np.random.default_rng(23)
    with env(CUDASW4_AMD_PIPE_TEST_DROP_STAGE=2, CUDASW4_AMD_PIPE_SPIN_LIMIT=2000, CUDASW4_AMD_PIPE_CPL=4):
        ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    q = rng.integers(0, 20, 200).astype(np.int8)
    seqs = [rng.integers(0, 21, 256 * 6).astype(np.int8), rng.integers(0, 21, 256 * 2).astype(np.int8)]
    t0 = time.time()
    scores, fails = run_pipeline(torch, capi, search, ctx, seqs, q, -11, -1, expect_fail=True)
    assert time.time() - t0 < 20.0
    assert fails >= 1
    assert scores[1] == -2.0                      # the six-stage subject lost its third stage
    chars, offsets, lengths = O.make_db(sorted(seqs, key=len))
    expect = O.scan(q, chars, offsets, lengths, simd=True)
    assert scores[0] in (-2.0, float(expect[0]))  # two stages: untouched by the drop, exact unless the abort reached it first


def test_watchdog_fails_a_scan_whose_side_launch_never_starts(monkeypatch):
    """the host side of the start handshake is bounded.  A side launch that is counted but never enqueued
    (test hook) leaves the bulk launch waiting in hipStreamWaitValue32 for a value nobody raises; collect() polls the scan's
    done event against a deadline, names the signal that never arrived, opens the gates by hand so that the device drains,
    and FAILS — within the deadline, not after a hang.  A fresh driver of the same process works."""
    from cudasw4_amd import driver, synthdb
    lengths = synthdb.sprot_like_lengths(30000, seed=5)
    chars, offsets, lengths = synthdb.random_db(lengths, seed=6)
    _, letters = O.read_fasta(FASTA)
    monkeypatch.setenv("CUDASW4_AMD_WATCHDOG_SECONDS", "2")
    monkeypatch.setenv("CUDASW4_AMD_TEST_LOSE_SIDE_LAUNCH", "1")
    # (the giants as a plain side launch, so that the launch that gets lost is one the bulk launch waits for)
    monkeypatch.setenv("CUDASW4_AMD_PIPELINES", "0")
    monkeypatch.setenv("CUDASW4_AMD_WINDOWS", "0")
    d = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2))
    d.db_from_arrays(chars, offsets, lengths)
    d.upload()
    if not d.handshake_active():
        d.close()
        pytest.skip("no start handshake on this runtime: nothing waits for a side launch")
    t0 = time.time()
    with pytest.raises(driver.DriverError) as err:
        d.scan(letters[12])
    took = time.time() - t0
    msg = str(err.value)
    assert "timed out" in msg and "side launch" in msg and "start signal" in msg, msg
    assert 2.0 <= took < 20.0, took
    d.close()
    monkeypatch.delenv("CUDASW4_AMD_TEST_LOSE_SIDE_LAUNCH")
    d = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2))
    d.db_from_arrays(chars, offsets, lengths)
    r = d.scan(letters[12])
    ids, sc = d.all_scores()
    pick = np.concatenate([np.arange(0, 30000, 601), [29999]])
    sub_chars = np.concatenate([chars[int(offsets[i]):int(offsets[i + 1])] for i in pick])
    sub_len = lengths[pick]
    sub_off = np.zeros(len(pick) + 1, np.uint64)
    sub_off[1:] = np.cumsum((sub_len.astype(np.int64) + 3) // 4 * 4)
    want = O.scan(O.encode(letters[12]), sub_chars, sub_off, sub_len, simd=True)
    by_id = np.empty(30000, np.int32)
    by_id[ids] = sc
    assert (by_id[pick] == want).all() and len(r["scores"]) == 10
    d.close()


def test_documented_binding_runs_at_the_host_drivers_rate():
    """the reference-side binding of INTEGRATION.md section 2 (the verbatim block: sw_set_query +
    sw_scan_batch + sw_batch_join + sw_topk) timed on a quarter-size peak DB and a quarter-size Swiss-Prot-like DB, next to
    the host driver on the same arrays: the same top scores, and a rate close to the driver's — at a quarter of the size the driver keeps two queries in flight,
    which a per-batch binding does not: 0.98 / 0.92 measured, asserted with a margin for run-to-run spread; at full size 1.00 /
    1.01 (profiles/r06_binding_bench.txt); the launcher-by-launcher form of the same document is reported beside it."""
    exe = os.path.join(ROOT, "tests", "boundary", "_build", "binding_gpu")
    if not os.path.exists(exe):
        pytest.skip("tests/boundary/_build/binding_gpu is built where /root/reference exists (__graft_entry__.build())")
    import importlib.util
    spec = importlib.util.spec_from_file_location("binding_bench", os.path.join(ROOT, "tools", "binding_bench.py"))
    bb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bb)
    rows = bb.bench(250_000, 142_500, 2)
    for name, batch, one_by_one, drv, ok in rows:
        assert ok, name
        assert batch >= 0.80 * drv, (name, batch, drv)   # measured 0.98 / 0.92; at full size 1.00 / 1.01


@pytest.mark.parametrize("extra,kernel,residency", [([], "half2", "resident"), (["--max-gpu-mem", "600M"], "half2", "hybrid"),
                                                    (["--kernel", "float"], "float", "resident"),
                                                    (["--kernel", "dpxs32"], "dpxs32", "resident"),
                                                    (["--kernel", "dpxs32", "--max-gpu-mem", "600M"], "dpxs32", "hybrid"),
                                                    (["--workload", "sprot-like"], "dpx", "resident")])
def test_bench_roofline_is_true_for_every_configuration(extra, kernel, residency):
    """The accounting behind `roofline` / `valu_roofline` (a streamed line reported frac 2.457): the DP
    kernels' busy time (union of the HIP-event intervals) fits the timed region, their own rate is at least the
    whole-job rate and below what the chip can issue, the traffic figure is scaled to the launch, and the VALU fraction
    — present whenever profiles/kernel_counters.json was measured on these kernel sources — lies in (0, 1] (round 6: also for the
    Swiss-Prot-like DB, whose giants' side launch was once taken for the dominant kernel and priced the fraction at 1.23)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    out = run_bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"] + extra)
    assert out["verified"] is True and out["config"]["kernel"] == kernel and out["config"]["residency"] == residency
    roof, valu = out["roofline"], out["valu_roofline"]
    assert 0 < valu["kernel_busy_ms_per_step"] <= out["ms_per_step"] * 1.001
    assert out["value"] <= valu["kernel_gcups"] * 1.001
    assert valu["kernel_gcups"] < (13500 if kernel in ("half2", "dpx") else 10500)
    assert 0 < roof["frac"] < 0.01 and roof["achieved"] > 0
    if residency == "hybrid":
        assert 0 < out["config"]["cached_chars"] < out["config"]["shard_chars"]
        assert roof["algorithmic_bytes_per_launch"] < 400e6     # a batch or the cached part, not the whole DB
    counters, _ = b.load_counters()
    if counters is not None:
        assert valu["frac"] is not None and 0 < valu["frac"] <= 1.0, valu
        if roof["traffic"] is not None:
            assert roof["traffic"] >= 0.9 * roof["algorithmic_bytes_per_launch"]
    else:
        assert valu["frac"] is None and "kernel_counters.json" in valu["counters_note"]
