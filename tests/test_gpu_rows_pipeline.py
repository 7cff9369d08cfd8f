"""GPU: very long subjects as pipelines of one-wave stages over many compute units (sw_scan_rows_pipelined,
csrc/sw_rows_pipeline.hpp) against the CPU oracle.

A subject is cut into spans of 64 x CPL columns, every span is a stage (one wave) that walks the query row by row and takes
the row's prefix maximum and its left neighbour's last H as one 64-bit word from device memory.  Bit-exact against the
oracle for every compiled width (4, 8, 16 columns per lane), for 1, 2 and many stages, for subjects that end anywhere
inside a lane's columns, a wave or right on a span border, for relatives of the query (long gapped alignments across lane
and stage borders), for every gap setting the prefix form allows, for queries whose length is / is not a multiple of the
hand-off batch, for several subjects of different stage counts in one launch.  (The failure path — a lost stage makes its
successors give up after a bounded number of polls, loudly — asserts on a wall clock: tests/test_gpu_zz_timing.py.)"""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import relatives

pytestmark = pytest.mark.gpu


def gpu_modules():
    import torch
    from cudasw4_amd import capi, search
    return torch, capi, search


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run_pipeline(torch, capi, search, ctx, seqs, q, gop, gex, maxlen=None, expect_fail=False):
    seqs = sorted(seqs, key=len)
    chars, offsets, lengths = O.make_db(seqs)
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    n = len(seqs)
    ctx.set_query(q)
    maxlen = int(maxlen if maxlen is not None else lengths.max())
    tb = ctx.scan_rows_pipelined_temp_bytes(n, maxlen)
    assert tb > 0
    temp = torch.empty(tb, dtype=torch.uint8, device="cuda")
    scores = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
    ids = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    fails = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.scan_rows_pipelined(db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(), 0, n, maxlen, gop, gex,
                            scores.data_ptr(), ids.data_ptr(), 1000, fails.data_ptr(), temp.data_ptr(), tb)
    torch.cuda.synchronize()
    if expect_fail:
        return scores.cpu().numpy(), int(fails.item())
    assert int(fails.item()) == 0
    expect = O.scan(q, chars, offsets, lengths, simd=True, gop=gop, gex=gex)
    np.testing.assert_array_equal(scores.cpu().numpy().astype(np.int64), expect.astype(np.int64))
    np.testing.assert_array_equal(ids.cpu().numpy(), 1000 + np.arange(n))


@pytest.mark.parametrize("cpl", [4, 8, 16])
def test_pipeline_every_width_and_stage_count_against_oracle(cpl):
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(300 + cpl)
    span = 64 * cpl
    with env(CUDASW4_AMD_PIPE_CPL=cpl):
        ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    q = rng.integers(0, 20, 333).astype(np.int8)
    # 1, 2 and many stages; ends right on / next to span, wave-row and lane borders
    lens = [1, 5, cpl, span - 1, span, span + 1, 2 * span, 2 * span - cpl + 1, 7 * span + 16 * cpl + 3, 23 * span, 23 * span - 1, 8001]
    seqs = [rng.integers(0, 21, int(l)).astype(np.int8) for l in lens]
    seqs += relatives(rng, q, 4, 2000, 23 * span)
    run_pipeline(torch, capi, search, ctx, seqs, q, -11, -1)
    run_pipeline(torch, capi, search, ctx, seqs[:6], q[:1], -11, -1)        # one query row
    run_pipeline(torch, capi, search, ctx, seqs[3:9], q[:16], -11, -1)      # exactly one hand-off batch
    run_pipeline(torch, capi, search, ctx, seqs[3:9], q[:17], -11, -1)      # ... and one row more
    run_pipeline(torch, capi, search, ctx, seqs[-3:], q, -11, -1, maxlen=40000)  # tickets of stages no subject has


@pytest.mark.parametrize("gop,gex", [(-11, -1), (-5, -5), (-20, -3), (-1, -1), (-40, 0)])
def test_pipeline_gap_scores_and_long_gapped_alignments(gop, gex):
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(17)
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    q = rng.integers(0, 20, 900).astype(np.int8)
    seqs = relatives(rng, q, 6, 8100, 20000) + [rng.integers(0, 21, 12000).astype(np.int8)]
    run_pipeline(torch, capi, search, ctx, seqs, q, gop, gex)


def test_pipeline_titin_sized_subject_and_long_query():
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(19)
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    q = rng.integers(0, 20, 5478).astype(np.int8)
    seqs = relatives(rng, q, 2, 30000, 35213) + [rng.integers(0, 21, 35213).astype(np.int8), rng.integers(0, 21, 9000).astype(np.int8)]
    run_pipeline(torch, capi, search, ctx, seqs, q, -11, -1)
    # a subject beyond the one-workgroup kernel's 40 960 residues
    seqs = [rng.integers(0, 21, 70001).astype(np.int8)] + relatives(rng, q[:700], 1, 60000, 70000)
    run_pipeline(torch, capi, search, ctx, seqs, q[:700], -11, -1)


def test_pipeline_argument_errors():
    torch, capi, search = gpu_modules()
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    ctx.set_query(np.zeros(10, dtype=np.int8))
    with pytest.raises(capi.SwError):
        ctx.scan_rows_pipelined(1, 1, 1, 0, 1, 100, -1, -11, 1, 1, 0, 0, 1, 1 << 20)       # gop > gex
    with pytest.raises(capi.SwError):
        ctx.scan_rows_pipelined(1, 1, 1, 0, 1, 100, -11, -1, 1, 1, 0, 0, 0, 0)             # no scratch
    with pytest.raises(capi.SwError):
        ctx.scan_rows_pipelined(1, 1, 1, 0, 1, 1 << 30, -11, -1, 1, 1, 0, 0, 1, 1 << 20)   # length x gex out of range
    ctx.scan_rows_pipelined(0, 0, 0, 0, 0, 100, -11, -1, 0, 0)                             # empty launch
    assert ctx.scan_rows_pipelined_temp_bytes(2, 35213) == 16 + 2 * 35 * 11 * 8     # 10 rows: 16 columns per lane, 35 stages (+ the launch's control words)


@pytest.mark.parametrize("kind", ["f32", "i32"])
def test_pipelined_rescore_takes_the_long_entries_of_an_overflow_list(kind):
    """sw_rescore_overflow_pipelined: the entries of a re-score list whose subject has at least min_subject_len residues are
    scored by pipeline stages and marked taken; sw_rescore_overflow_claim behind it scores the rest.  Together: every listed
    subject's exact score (== oracle), nobody scored twice (entries all taken afterwards), unlisted subjects untouched, the
    reference's overflow statistic counts every listed subject at or above the packed limit exactly once."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(41)
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    q = rng.integers(0, 20, 700).astype(np.int8)
    seqs = [rng.integers(0, 21, int(l)).astype(np.int8) for l in rng.integers(40, 900, 60)]
    seqs += relatives(rng, q, 8, 1300, 6000) + [rng.integers(0, 21, int(l)).astype(np.int8) for l in (2500, 4100, 7999)]
    seqs = sorted(seqs, key=len)
    chars, offsets, lengths = O.make_db(seqs)
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    n = len(seqs)
    ctx.set_query(q)
    expect = O.scan(q, chars, offsets, lengths, simd=True)
    listed = np.array(sorted(set(rng.choice(n, 30, replace=False).tolist() + list(range(n - 11, n)))), dtype=np.int32)
    rng.shuffle(listed)
    cap = n
    lst = torch.full((cap,), 12345, dtype=torch.int32, device="cuda")     # entries behind the count are never read
    lst[:len(listed)] = torch.from_numpy(listed).cuda()
    count = torch.tensor([len(listed)], dtype=torch.int32, device="cuda")
    scores = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
    ids = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    fails = torch.zeros(1, dtype=torch.int32, device="cuda")
    over = torch.zeros(1, dtype=torch.int32, device="cuda")
    maxlen = int(lengths.max())
    tb = ctx.rescore_overflow_pipelined_temp_bytes(maxlen)
    k = capi.KIND_F32 if kind == "f32" else capi.KIND_I32
    tb2 = ctx.scan_temp_bytes(k, -1, cap, maxlen)
    temp = torch.empty(max(tb, tb2, 256), dtype=torch.uint8, device="cuda")
    limit = 2048
    ctx.rescore_overflow_pipelined(lst.data_ptr(), count.data_ptr(), cap, db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(),
                                   maxlen, 1300, -11, -1, scores.data_ptr(), ids.data_ptr(), 500, fails.data_ptr(), limit, over.data_ptr(),
                                   temp.data_ptr(), temp.numel())
    torch.cuda.synchronize()
    after_pick = lst.cpu().numpy()[:len(listed)]
    long_listed = lengths[listed] >= 1300
    assert (after_pick[long_listed] == -2).all() and (after_pick[~long_listed] == listed[~long_listed]).all()
    got = scores.cpu().numpy()
    assert (got[listed[long_listed]] == expect[listed[long_listed]]).all() and (got[listed[~long_listed]] == -7).all()
    ctx.rescore_overflow_claim(k, lst.data_ptr(), count.data_ptr(), cap, db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(),
                               maxlen, -11, -1, scores.data_ptr(), ids.data_ptr(), 500, temp.data_ptr(), temp.numel(), limit, over.data_ptr())
    torch.cuda.synchronize()
    assert int(fails.item()) == 0
    got, gid = scores.cpu().numpy(), ids.cpu().numpy()
    mask = np.zeros(n, dtype=bool)
    mask[listed] = True
    assert (got[mask] == expect[mask]).all() and (got[~mask] == -7).all()
    assert (gid[mask] == 500 + np.nonzero(mask)[0]).all() and (gid[~mask] == -7).all()
    assert (lst.cpu().numpy()[:len(listed)] == -2).all()
    assert int(over.item()) == int((expect[mask] >= limit).sum()) > 0
