"""CPU: the algebra of the row-parallel kernels (cudasw4_amd/csrc/sw_rows_pipeline.hpp) restated in numpy and checked against
the oracle's scalar DP.

The kernel walks the query row by row and resolves the horizontal gap of a whole row as ONE max-plus prefix:
    E(i,j) = gop + (j-1) gex + max_{k<j} ( H~(i,k) - k gex ),    H~ = max(0, H(i-1,j-1) + s, F(i,j))   (H without E)
which is exact for gop <= gex.  This file pins that identity (and the claim that columns behind the subject's end, scored
-30000 against everything, can stay unmasked) without a GPU; tests/test_gpu_rows_pipeline.py checks the kernel itself."""
import numpy as np
import pytest

import oracle_lib as O


def rows_score(q, s, m21, gop, gex, pad=0):
    """the kernel's arithmetic, one numpy vector per query row; `pad` extra columns behind the subject's end"""
    L = len(s) + pad
    letters = np.concatenate([s.astype(np.int64), np.full(pad, 21, dtype=np.int64)])
    sub = np.concatenate([m21.reshape(-1, 21).astype(np.int64), np.full((m21.size // 21, 1), -30000, dtype=np.int64)], axis=1)
    k = np.arange(1, L + 1, dtype=np.int64)          # 1-based column
    H = np.zeros(L, dtype=np.int64)
    F = np.full(L, -10000, dtype=np.int64)
    best = 0
    for qi in q:
        F = np.maximum(F + gex, H + gop)
        diag = np.concatenate([[0], H[:-1]])
        ht = np.maximum(np.maximum(diag + sub[int(qi)][letters], F), 0)
        G = ht - k * gex
        pref = np.concatenate([[-(1 << 40)], np.maximum.accumulate(G)[:-1]])   # exclusive prefix maximum
        E = gop + (k - 1) * gex + pref
        H = np.maximum(ht, E)
        best = max(best, int(H.max()))               # (including the columns behind the end: they never exceed the rest)
    return best


@pytest.mark.parametrize("gop,gex", [(-11, -1), (-5, -5), (-20, -3), (-1, -1), (-40, 0)])
def test_prefix_form_of_the_horizontal_gap_is_exact(gop, gex):
    rng = np.random.default_rng(abs(gop) * 100 + abs(gex))
    m21 = O.blosum21(62)
    for trial in range(12):
        qlen, slen = int(rng.integers(1, 90)), int(rng.integers(1, 400))
        q = rng.integers(0, 20, qlen).astype(np.int8)
        s = rng.integers(0, 21, slen).astype(np.int8)
        if trial % 2:  # a relative of the query inside the subject: long gapped alignments
            copy = [int(c) for c in q for _ in range(1 if rng.random() > 0.1 else 0)]
            for _ in range(3):
                at = int(rng.integers(0, max(1, len(copy))))
                copy[at:at] = rng.integers(0, 20, int(rng.integers(1, 9))).tolist()
            copy = np.array(copy[:slen], dtype=np.int8)
            at = int(rng.integers(0, slen - len(copy) + 1))
            s[at:at + len(copy)] = copy
        chars, offsets, lengths = O.make_db([s])
        want = int(O.scan(q, chars, offsets, lengths, gop=gop, gex=gex)[0])
        assert rows_score(q, s, m21, gop, gex) == want, (trial, qlen, slen)
        assert rows_score(q, s, m21, gop, gex, pad=int(rng.integers(1, 70))) == want, ("padded", trial)


def test_prefix_form_needs_gop_not_above_gex():
    """with gop > gex the identity fails (opening twice beats extending): the library refuses such scores for sw_scan_rows_pipelined"""
    rng = np.random.default_rng(3)
    m21 = O.blosum21(62)
    differs = 0
    for _ in range(40):
        q = rng.integers(0, 20, 40).astype(np.int8)
        s = np.concatenate([q[:20], rng.integers(0, 20, 6).astype(np.int8), q[20:]])
        chars, offsets, lengths = O.make_db([s])
        differs += rows_score(q, s, m21, -1, -6) != int(O.scan(q, chars, offsets, lengths, gop=-1, gex=-6)[0])
    assert differs > 0


# ------------------------------------------------------------------------------------------------------------------
# The pipelined form (cudasw4_amd/csrc/sw_rows_pipeline.hpp): the subject is cut into spans, every span is a stage that
# walks ALL query rows and takes two numbers per row from the stage to its left — carry(i), the prefix maximum of
# H~(i,k) - k gex over all columns left of the span, and hlast(i), the H of the neighbour's last column (next row's
# diagonal input).  Restated stage by stage (a stage runs to its end before the next one starts: what the hand-off array
# allows), with the kernel's frames: lanes of `cpl` columns, lane-local G = H~ - c gex, kg0 = (col0 + 1) gex.
def pipeline_score(q, s, m21, gop, gex, span, cpl):
    NEG = -(1 << 29)
    L = len(s)
    nstages = max(1, -(-L // span))
    sub = np.concatenate([m21.reshape(-1, 21).astype(np.int64), np.full((m21.size // 21, 1), -30000, dtype=np.int64)], axis=1)
    carry_in = np.full(len(q), NEG, dtype=np.int64)   # what stage 0 "receives"
    hlast_in = np.zeros(len(q), dtype=np.int64)
    best_in = 0
    for st in range(nstages):
        cols = np.arange(st * span, (st + 1) * span)                       # 0-based owned columns (past the end: letter 21)
        letters = np.where(cols < L, s[np.minimum(cols, L - 1)].astype(np.int64), 21)
        lane_of = (cols - st * span) // cpl
        c_of = (cols - st * span) % cpl
        kg0 = (st * span + lane_of * cpl + 1) * gex
        H = np.zeros(span, dtype=np.int64)
        F = np.full(span, -10000, dtype=np.int64)
        hleft = 0
        best = 0
        carry_out = np.empty(len(q), dtype=np.int64)
        hlast_out = np.empty(len(q), dtype=np.int64)
        for i, qi in enumerate(q):
            F = np.maximum(F + gex, H + gop)
            diag = np.concatenate([[hleft], H[:-1]])
            ht = np.maximum(np.maximum(diag + sub[int(qi)][letters], F), 0)
            G = ht - c_of * gex                                            # lane-local frame
            Gg = G - kg0                                                   # global frame: H~ - k gex
            inc_all = max(int(Gg.max()), int(carry_in[i]))
            pref = np.concatenate([[carry_in[i]], np.maximum(np.maximum.accumulate(Gg)[:-1], carry_in[i])])
            E = pref + kg0 + gop + (c_of - 1) * gex
            H = np.maximum(ht, E)
            best = max(best, int(H.max()))
            carry_out[i] = inc_all
            hlast_out[i] = H[-1]
            hleft = int(hlast_in[i])                                       # H(i, col0 - 1): next row's diagonal input
        carry_in, hlast_in = carry_out, hlast_out
        best_in = max(best_in, best)
    return best_in


@pytest.mark.parametrize("gop,gex", [(-11, -1), (-5, -5), (-20, -3), (-40, 0)])
def test_pipelined_stages_hand_over_prefix_and_last_column(gop, gex):
    rng = np.random.default_rng(50 + abs(gop) + abs(gex))
    m21 = O.blosum21(62)
    for trial in range(10):
        qlen, slen = int(rng.integers(1, 70)), int(rng.integers(1, 600))
        q = rng.integers(0, 20, qlen).astype(np.int8)
        s = rng.integers(0, 21, slen).astype(np.int8)
        if trial % 2:  # a relative of the query across span borders
            copy = [int(c) for c in q for _ in range(1 if rng.random() > 0.1 else 0)]
            for _ in range(3):
                at = int(rng.integers(0, max(1, len(copy))))
                copy[at:at] = rng.integers(0, 20, int(rng.integers(1, 9))).tolist()
            copy = np.array(copy[:slen], dtype=np.int8)
            at = int(rng.integers(0, slen - len(copy) + 1))
            s[at:at + len(copy)] = copy
        chars, offsets, lengths = O.make_db([s])
        want = int(O.scan(q, chars, offsets, lengths, gop=gop, gex=gex)[0])
        for span, cpl in ((16, 4), (64, 8), (32, 16)):
            assert pipeline_score(q, s, m21, gop, gex, span, cpl) == want, (trial, qlen, slen, span, cpl)
