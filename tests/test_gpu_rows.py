"""GPU: the row-parallel scan of very long subjects (sw_scan_rows, csrc/sw_rows_kernel.hpp) against the CPU oracle.

One 1024-thread workgroup per subject walks the query row by row, the horizontal gap as a max-plus prefix over the
workgroup.  Bit-exact against the oracle's scalar / SIMD DP for every compiled width (8 ... 40 columns per thread), for
subjects that end anywhere inside a thread's columns or a wave, for relatives of the query (long gapped alignments that
cross thread and wave borders), for gap scores with gop == gex, and for queries longer than the 4096-letter LDS chunk."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def gpu_modules():
    import torch
    from cudasw4_amd import capi, search
    return torch, capi, search


def relatives(rng, q, n, lo, hi):
    """subjects that contain mutated copies of the query (substitutions, insertions, deletions) inside random flanks"""
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        s = rng.integers(0, 20, L).astype(np.int8)
        copy = []
        for c in q:
            r = rng.random()
            if r < 0.04:
                continue                                   # deletion
            if r < 0.08:
                copy.extend(rng.integers(0, 20, int(rng.integers(1, 12))).tolist())   # insertion
            copy.append(int(rng.integers(0, 20)) if r > 0.85 else int(c))
        copy = np.array(copy[:L], dtype=np.int8)
        at = int(rng.integers(0, L - len(copy) + 1))
        s[at:at + len(copy)] = copy
        out.append(s)
    return out


def run_rows(torch, capi, search, ctx, seqs, q, gop, gex, maxlen=None):
    seqs = sorted(seqs, key=len)
    chars, offsets, lengths = O.make_db(seqs)
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    n = len(seqs)
    ctx.set_query(q)
    scores = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
    ids = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    ctx.scan_rows(db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(), 0, n,
                  int(maxlen if maxlen is not None else lengths.max()), gop, gex, scores.data_ptr(), ids.data_ptr(), 1000)
    torch.cuda.synchronize()
    expect = O.scan(q, chars, offsets, lengths, simd=True, gop=gop, gex=gex)
    np.testing.assert_array_equal(scores.cpu().numpy().astype(np.int64), expect.astype(np.int64))
    np.testing.assert_array_equal(ids.cpu().numpy(), 1000 + np.arange(n))


@pytest.mark.parametrize("width", [8, 16, 24, 32, 36, 40])
def test_rows_every_width_against_oracle(width):
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(100 + width)
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    top = 1024 * width
    q = rng.integers(0, 20, 333).astype(np.int8)
    lens = [top, top - 1, top - width, top - 64 * width + 3, max(1, 1024 * (width - 8) + 1), 8001, 5, 1]
    seqs = [rng.integers(0, 21, int(l)).astype(np.int8) for l in lens]
    seqs += relatives(rng, q, 4, max(2000, top // 2), top)
    run_rows(torch, capi, search, ctx, seqs, q, -11, -1, maxlen=top)
    run_rows(torch, capi, search, ctx, seqs[:3], q[:1], -11, -1, maxlen=top)       # one query row


@pytest.mark.parametrize("gop,gex", [(-11, -1), (-5, -5), (-20, -3), (-1, -1), (-40, 0)])
def test_rows_gap_scores_and_long_gapped_alignments(gop, gex):
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(7)
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    q = rng.integers(0, 20, 900).astype(np.int8)
    seqs = relatives(rng, q, 6, 8100, 20000) + [rng.integers(0, 21, 12000).astype(np.int8)]
    run_rows(torch, capi, search, ctx, seqs, q, gop, gex)


def test_rows_long_query_crosses_the_lds_chunk():
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(9)
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    q = rng.integers(0, 20, 4096 + 1500).astype(np.int8)
    seqs = relatives(rng, q, 2, 9000, 12000) + [rng.integers(0, 21, 8500).astype(np.int8)]
    run_rows(torch, capi, search, ctx, seqs, q, -11, -1)


def test_rows_argument_errors():
    torch, capi, search = gpu_modules()
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    ctx.set_query(np.zeros(10, dtype=np.int8))
    assert capi.scan_rows_max_subject() == 40960
    with pytest.raises(capi.SwError):
        ctx.scan_rows(1, 1, 1, 0, 1, 100, -1, -11, 1, 1)          # gop > gex
    with pytest.raises(capi.SwError):
        ctx.scan_rows(1, 1, 1, 0, 1, 40961, -11, -1, 1, 1)        # subject bound above the limit
    ctx.scan_rows(0, 0, 0, 0, 0, 100, -11, -1, 0, 0)              # empty launch
