"""GPU parity of the STREAMED packed kernels (sw_stream_kernel.hpp): rounds of several slots per alignment group, zero-level
jumps at the slot borders, separator columns, letter words that straddle two subjects, slot maxima carried over the stripes,
successors of high-scoring subjects flagged and re-scored.  Every score against the CPU oracle, through the C ABI.

The launches of a test DB have fewer batches than the GPU has workgroup slots, so a workgroup would claim one batch at a
time; CUDASW4_AMD_GRID_CAP (a test hook of the library) caps the persistent grid, which makes the claims long."""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import gpu_modules, scan_all_scores, kinds_configs

pytestmark = pytest.mark.gpu


class env:
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def ragged(rng, lens, other_every=7):
    return [rng.integers(0, 21 if i % other_every == 0 else 20, int(l)).astype(np.int8) for i, l in enumerate(lens)]


def packed_configs(search, capi):
    c = kinds_configs(search, capi)
    return {k: c[k] for k in ("half2+float", "dpxs16+dpxs32")}


@pytest.mark.parametrize("cap", [1, 3])
def test_streamed_rounds_match_the_oracle_on_ragged_subjects(cap):
    """2 500 ragged subjects (1 ... 600 residues, letters 0..20) on a grid of 1 / 3 workgroups: rounds of up to 16 slots,
    slot widths that are no multiples of four (letter words straddle slot borders), slots shorter than the minimum width,
    the last round partial; single-stripe (R = 17 ... 48) and multi-stripe queries (2 ... 4 stripes), both packed kinds."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(11 + cap)
    lens = np.concatenate([rng.integers(1, 40, 300), rng.integers(40, 300, 1500), rng.integers(300, 600, 700)])
    seqs = ragged(rng, lens)
    db = O.make_db(seqs)
    with env(CUDASW4_AMD_GRID_CAP=cap):
        for ql in (257, 300, 511, 768, 769, 1000, 1600, 2100):
            q = rng.integers(0, 20, ql).astype(np.int8)
            expect = O.scan(q, *db, simd=True)
            for cfg, kt in packed_configs(search, capi).items():
                got, res, _ = scan_all_scores(search, capi, db, q, kernel_types=kt)
                np.testing.assert_array_equal(got, expect, err_msg="%s q=%d cap=%d" % (cfg, ql, cap))


def test_streamed_equals_one_batch_at_a_time():
    """CUDASW4_AMD_STREAM=1 is sw_scan_kernel (one batch at a time), 4 and 16 the streamed kernel: same scores."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(5)
    seqs = ragged(rng, rng.integers(30, 500, 3000))
    db = O.make_db(seqs)
    q = rng.integers(0, 20, 900).astype(np.int8)
    expect = O.scan(q, *db, simd=True)
    for stream in (1, 4, 16):
        with env(CUDASW4_AMD_GRID_CAP=2, CUDASW4_AMD_STREAM=stream):
            for cfg, kt in packed_configs(search, capi).items():
                got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt)
                np.testing.assert_array_equal(got, expect, err_msg="%s stream=%d" % (cfg, stream))


def test_high_scoring_subjects_flag_their_successors():
    """Relatives of the query (scores far above the jump of the zero levels) scattered over the DB, also back to back and at
    the end of a round: every score still equals the oracle — the subjects that follow a hit in a lane's stream are
    re-scored by the 32-bit kind — and the re-scored count says the flags fired."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(7)
    ql = 640
    q = rng.integers(0, 20, ql).astype(np.int8)
    lens = np.sort(rng.integers(150, 641, 4000))
    seqs = ragged(rng, lens, other_every=10 ** 9)
    planted = sorted(set(rng.integers(0, 4000, 120).tolist() + [64, 65, 66, 96, 3999, 3998, 0]))
    for i in planted:
        L = len(seqs[i])
        b = int(rng.integers(0, ql - min(L, ql) + 1))
        m = q[b:b + L].copy()
        mut = rng.random(len(m)) < rng.choice([0.05, 0.25])   # (scores above the fp16 jump of 512 / above the int16 jump of 2048)
        m[mut] = rng.integers(0, 20, int(mut.sum()))
        seqs[i] = np.concatenate([m, rng.integers(0, 20, L - len(m)).astype(np.int8)])[:L]
    db = O.make_db(seqs)
    expect = O.scan(q, *db, simd=True)
    assert (expect >= 600).sum() >= 60 and (expect >= 2100).sum() >= 10
    for cap in (1, 4):
        with env(CUDASW4_AMD_GRID_CAP=cap):
            for cfg, kt in packed_configs(search, capi).items():
                got, res, _ = scan_all_scores(search, capi, db, q, kernel_types=kt)
                np.testing.assert_array_equal(got, expect, err_msg="%s cap=%d" % (cfg, cap))
            # multi-stripe query, same relatives (a prefix of the longer query is the old one)
            q2 = np.concatenate([q, rng.integers(0, 20, 700).astype(np.int8)])
            expect2 = O.scan(q2, *db, simd=True)
            for cfg, kt in packed_configs(search, capi).items():
                got, res, _ = scan_all_scores(search, capi, db, q2, kernel_types=kt)
                np.testing.assert_array_equal(got, expect2, err_msg="%s cap=%d multi" % (cfg, cap))


def test_streamed_rounds_with_other_gap_scores_and_matrices():
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(3)
    seqs = ragged(rng, np.sort(rng.integers(20, 350, 2200)))
    q1 = rng.integers(0, 20, 420).astype(np.int8)
    q2 = rng.integers(0, 20, 1300).astype(np.int8)
    seqs[100] = q1[40:300].copy()
    seqs[2000] = q2[:340].copy()
    db = O.make_db(seqs)
    with env(CUDASW4_AMD_GRID_CAP=2):
        for which, gop, gex in ((45, -13, -3), (80, -5, -2), (50, -10, -4), (62, -3, -3)):
            m = O.blosum21(which)
            for q in (q1, q2):
                expect = O.scan(q, *db, m21=m, gop=gop, gex=gex, simd=True)
                for cfg, kt in packed_configs(search, capi).items():
                    got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt, gop=gop, gex=gex, matrix=m)
                    np.testing.assert_array_equal(got, expect, err_msg="%s q=%d blosum%d %d/%d" % (cfg, len(q), which, gop, gex))


def test_long_subjects_take_one_slot_rounds_and_lower_their_frame():
    """Subjects beyond the column budget of a round (1 300 ... 7 000 residues on 16-lane groups: partition 34 with enough
    subjects) run as rounds of one slot whose frame is lowered every K columns, between rounds of many short subjects."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(9)
    lens = np.sort(np.concatenate([rng.integers(60, 200, 1500), rng.integers(1300, 7000, 600)]))
    seqs = ragged(rng, lens, other_every=13)
    q = rng.integers(0, 20, 1100).astype(np.int8)
    seqs[-1] = np.concatenate([seqs[-1][:3000], q, seqs[-1][3000:]])[:len(seqs[-1])]
    db = O.make_db(seqs)
    expect = O.scan(q, *db, simd=True)
    with env(CUDASW4_AMD_GRID_CAP=2):
        for cfg, kt in packed_configs(search, capi).items():
            got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt)
            np.testing.assert_array_equal(got, expect, err_msg=cfg)


def test_uniform_db_every_slot_count():
    """Identical subjects (the peak benchmark's shape), lengths around the word and block borders, every cap of slots."""
    torch, capi, search = gpu_modules()
    _, qs = O.load_queries()
    for L in (31, 127, 128, 129, 131, 255):
        codes = O.pseudodb_codes(L, 42)
        db = search.DeviceDB.pseudo(1500, L, codes, device=0)
        for qi in (4, 9, 13):
            want = O.scan(qs[qi], *O.make_db([codes]), simd=True)[0]
            for slots in (2, 3, 7, 16):
                with env(CUDASW4_AMD_GRID_CAP=1, CUDASW4_AMD_STREAM=slots):
                    for cfg, kt in packed_configs(search, capi).items():
                        s = search.Searcher(device=0, num_top=0, matrix=O.blosum21(62), kernel_types=kt)
                        s.set_database(db)
                        s.scan(qs[qi])
                        sc = s.all_scores()
                        assert sc.min() == sc.max() == want, (cfg, L, qi, slots, int(sc.min()), int(sc.max()), int(want))
