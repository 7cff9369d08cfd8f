"""GPU parity: the HIP path (through the C ABI) against the oracle / the reference's golden vectors.

Bar: bit-exact integer scores for every kind (the half2/float kinds hold integers exactly below
2048 / 2^24 and hand anything larger to the 32-bit re-score, so they are bit-exact too)."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_util import gpu_modules, scan_all_scores, kinds_configs

pytestmark = pytest.mark.gpu


def check_overflow_count(num_overflows, expect, lengths, limit):
    """Every subject whose score reaches the limit of the packed kind must have been re-scored.  The column-offset
    kernels keep all values of column j raised by a*(j + lanes) and flag a subject already when the bound
    score + a*(columns of its wave + 2*lanes) reaches the limit; the launcher uses them only while a*columns stays
    leaves room (1536 of 2048, 12500 of 25000), so nothing scoring below limit - room is ever flagged."""
    expect = np.asarray(expect)
    room = {2048: 1536, 25000: 12500}[limit]
    lo = int((expect >= limit).sum())
    hi = int((expect >= limit - room).sum())
    assert lo <= num_overflows <= hi, (num_overflows, lo, hi)


def test_library_sees_gpu():
    torch, capi, search = gpu_modules()
    assert capi.device_count() >= 1
    ctx = capi.Context(0)
    ctx.close()


@pytest.mark.parametrize("cfg", ["half2+float", "dpxs16+dpxs32", "dpxs32", "float"])
def test_golden_pairs_all_kinds(cfg):
    torch, capi, search = gpu_modules()
    g = O.golden("ref_scores.json")
    kt = kinds_configs(search, capi)[cfg]
    for p in g["pairs"]:
        q = np.array(p["q"], dtype=np.int8)
        s = np.array(p["s"], dtype=np.int8)
        db = O.make_db([s])
        got, res, _ = scan_all_scores(search, capi, db, q, kernel_types=kt, gop=p.get("gop", -11), gex=p.get("gex", -1))
        assert got.tolist() == [p["score"]], (cfg, len(q), len(s), res.num_overflows)


@pytest.mark.parametrize("cfg", ["half2+float", "dpxs16+dpxs32", "dpxs32", "float"])
@pytest.mark.parametrize("merge", [True, False])
def test_allvsall_matches_reference_dp(cfg, merge):
    """20 x 20 of allqueries.fasta: single-stripe and multi-stripe queries, packed overflow
    (>= 2048 fp16, >= 25000 int16) with 32-bit re-score, partitions 6..34."""
    torch, capi, search = gpu_modules()
    g = O.golden("ref_scores.json")
    _, qs = O.load_queries()
    db = O.make_db(qs)
    kt = kinds_configs(search, capi)[cfg]
    expect = np.array(g["allvsall"], dtype=np.int32)
    lens = np.array([len(x) for x in qs])
    for i, q in enumerate(qs):
        got, res, _ = scan_all_scores(search, capi, db, q, kernel_types=kt, merge=merge)
        assert got.tolist() == expect[i].tolist(), (cfg, i)
        if cfg == "half2+float":
            check_overflow_count(res.num_overflows, expect[i], lens, 2048)
        if cfg == "dpxs16+dpxs32":
            check_overflow_count(res.num_overflows, expect[i], lens, 25000)


@pytest.mark.parametrize("cfg", ["half2+float", "dpxs16+dpxs32", "dpxs32", "float"])
def test_pseudo_db_matches_reference_dp(cfg):
    torch, capi, search = gpu_modules()
    g = O.golden("ref_scores.json")
    _, qs = O.load_queries()
    kt = kinds_configs(search, capi)[cfg]
    s = search.Searcher(device=0, num_top=0, matrix=O.blosum21(62), kernel_types=kt)
    for L, expect in g["pseudo"].items():
        L = int(L)
        db = search.DeviceDB.pseudo(999, L, O.pseudodb_codes(L, 42), device=0)  # odd count: tail handling
        s.set_database(db)
        for qi, q in enumerate(qs):
            s.scan(q)
            sc = s.all_scores()
            assert sc.min() == sc.max() == expect[qi], (cfg, L, qi)


def test_long_subject_partition35():
    torch, capi, search = gpu_modules()
    g = O.golden("ref_scores.json")["long_subject"]
    _, qs = O.load_queries()
    subj = np.concatenate([qs[i] for i in g["concat_of_queries"]])
    db = O.make_db([qs[0], subj, qs[3]])
    for cfg, kt in kinds_configs(search, capi).items():
        for qi in (0, 7, 12, 19):
            got, _, _ = scan_all_scores(search, capi, db, qs[qi], kernel_types=kt)
            assert got[1] == g["scores"][qi], (cfg, qi)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_ragged_db_vs_oracle(seed):
    """Ragged lengths incl. 1-residue and > 1280 subjects, unknown letters (code 20), odd subject
    count, query lengths around the stripe borders, other matrices and gap scores."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(seed)
    n = 777
    lens = np.concatenate([rng.integers(1, 60, 200), rng.integers(60, 700, 500), rng.integers(700, 3000, 77)])
    seqs = [rng.integers(0, 21 if i % 5 == 0 else 20, int(l)).astype(np.int8) for i, l in enumerate(lens)]
    # plant homologs of the query so that scores are not all tiny
    qlen = [16, 64, 65, 191, 192, 193, 500, 512, 513, 1025, 1500][seed * 3:seed * 3 + 5]
    db = O.make_db(seqs)
    assert len(seqs) == n
    for ql in qlen:
        q = rng.integers(0, 20, ql).astype(np.int8)
        seqs2 = list(seqs)
        seqs2[5] = np.concatenate([rng.integers(0, 20, 30).astype(np.int8), q, rng.integers(0, 20, 11).astype(np.int8)])
        seqs2[6] = q[: max(1, ql // 2)].copy()
        db = O.make_db(seqs2)
        for which, gop, gex in ((62, -11, -1), (45, -13, -3), (80, -5, -2)):
            m = O.blosum21(which)
            expect = O.scan(q, *db, m21=m, gop=gop, gex=gex, simd=True)
            for cfg, kt in kinds_configs(search, capi).items():
                got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt, gop=gop, gex=gex, matrix=m)
                np.testing.assert_array_equal(got, expect, err_msg="%s q=%d blosum%d" % (cfg, ql, which))


def test_empty_and_tiny_inputs():
    torch, capi, search = gpu_modules()
    q = np.array([3], dtype=np.int8)
    db = O.make_db([np.array([3], dtype=np.int8)])
    for cfg, kt in kinds_configs(search, capi).items():
        got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt)
        assert got.tolist() == [6]  # D-D in BLOSUM62
    # n == 0 is a no-op through the ABI
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    ctx.set_query(q)
    ctx.scan_partition(capi.KIND_I16X2, 0, 0, 0, 0, 0, 0, 0, -11, -1, 0, 0)
    with pytest.raises(capi.SwError):
        ctx.scan_partition(capi.KIND_I16X2, 0, 0, 0, 0, 0, 5, 10, +1, -1, 0, 0)  # positive gap score
    with pytest.raises(capi.SwError) as ei:   # the last partition's nominal boundary (INT_MAX) is not a subject length
        ctx.scan_partition(capi.KIND_I16X2, 35, 1, 1, 1, 0, 5, 2**31 - 1, -11, -1, 1, 1)
    assert "longest subject of the range" in str(ei.value)
    with pytest.raises(capi.SwError):
        ctx.set_query(np.zeros(0, dtype=np.int8))
    ctx2 = capi.Context(0)
    with pytest.raises(capi.SwError):
        ctx2.scan_partition(capi.KIND_I16X2, 0, 1, 1, 1, 0, 5, 10, -11, -1, 1, 1)  # no matrix / query yet


def test_topk_matches_oracle_order():
    torch, capi, search = gpu_modules()
    _, qs = O.load_queries()
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, 20, int(l)).astype(np.int8) for l in np.sort(rng.integers(20, 400, 5000))]
    db_arrays = O.make_db(seqs)
    q = qs[2]
    got_all, res, order = scan_all_scores(search, capi, db_arrays, q, num_top=25)
    # ids refer to the sorted DB; tie order = ascending id
    sorted_scores = got_all[order]
    es, ei = O.topk(sorted_scores, 25)
    assert res.scores.tolist() == es.tolist()
    assert res.reference_ids.tolist() == ei.tolist()


@pytest.mark.parametrize("n,k,hi", [(200_000, 10, 40), (200_000, 1000, 3000), (300_001, 25, 2), (1_000_000, 10, 1),
                                    (150_000, 7, 1 << 20), (131_072, 16_384, 50), (3_000_001, 32, 3), (700, 5, 9), (2049, 1, 100)])
def test_topk_select_equals_sort_and_oracle(n, k, hi, monkeypatch):
    """sw_topk's radix select (large n) against its full-sort path and the oracle's top-K: heavy ties (few distinct
    scores: the ties at the threshold are broken by position), all-equal scores, nearly distinct scores, a large k."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(n + k)
    scores_i = rng.integers(0, hi, n).astype(np.int32)
    es, ei = O.topk(scores_i, k)
    ctx = capi.Context(0)
    d_s = torch.from_numpy(scores_i.astype(np.float32)).cuda()
    d_i = torch.arange(1000, 1000 + n, dtype=torch.int32, device="cuda")  # ids need not equal positions
    got = {}
    for path in ("sort", "select") + (("small",) if k <= 32 else ()):   # (round 6: two launches for k <= 32, sw_api.hip: topk_small_*)
        monkeypatch.setenv("CUDASW4_AMD_TOPK", path)
        tb = capi.topk_temp_bytes(n, k)
        temp = torch.empty(tb, dtype=torch.uint8, device="cuda")
        out_s = torch.full((k,), -7.0, dtype=torch.float32, device="cuda")
        out_i = torch.full((k,), -7, dtype=torch.int32, device="cuda")
        ctx.topk(d_s.data_ptr(), d_i.data_ptr(), n, k, out_s.data_ptr(), out_i.data_ptr(), temp.data_ptr(), tb, 0)
        torch.cuda.synchronize()
        got[path] = (out_s.cpu().numpy().astype(np.int64).tolist(), (out_i.cpu().numpy().astype(np.int64) - 1000).tolist())
    assert got["select"] == got["sort"] and all(v == got["sort"] for v in got.values())
    assert got["select"][0] == es.tolist() and got["select"][1] == ei.tolist()
    ctx.close()


def test_full_size_peak_db_property():
    """BASELINE config 2 at full size (10^6 x 512): every subject is the same sequence, so every
    one of the 10^6 scores must equal the reference-DP golden score; a checksum covers all slots."""
    torch, capi, search = gpu_modules()
    g = O.golden("ref_scores.json")
    _, qs = O.load_queries()
    L, num = 512, 1_000_000
    db = search.DeviceDB.pseudo(num, L, O.pseudodb_codes(L, 42), device=0)
    for cfg in ("half2+float", "dpxs16+dpxs32"):
        s = search.Searcher(device=0, num_top=10, matrix=O.blosum21(62), kernel_types=kinds_configs(search, capi)[cfg])
        s.set_database(db)
        for qi in (0, 9, 19):
            res = s.scan(qs[qi])
            exp = g["pseudo"]["512"][qi]
            sc = s.scores[:num]
            assert float(sc.min().item()) == float(sc.max().item()) == float(exp)
            assert int(s.ids[:num].to(torch.int64).sum().item()) == num * (num - 1) // 2
            assert res.scores.tolist() == [exp] * 10 and res.reference_ids.tolist() == list(range(10))
            assert res.num_overflows == 0


def test_temp_sizing_and_too_small_temp():
    """sw_scan_temp_bytes is 0 for single-stripe queries, > 0 for longer ones; a temp buffer smaller than
    one workgroup's border scratch is refused (SW_ERR_TEMP), a smaller-than-ideal one only shrinks the grid."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(4)
    seqs = [rng.integers(0, 20, int(l)).astype(np.int8) for l in np.sort(rng.integers(100, 600, 4000))]
    chars, offsets, lengths = O.make_db(seqs)
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    n = len(seqs)
    scores = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
    ids = torch.zeros(n, dtype=torch.int32, device="cuda")
    ovf_pos = torch.zeros(n, dtype=torch.int32, device="cuda")
    ovf_cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    q_short = rng.integers(0, 20, 300).astype(np.int8)
    q_long = rng.integers(0, 20, 1300).astype(np.int8)
    ctx.set_query(q_short)
    assert ctx.scan_temp_bytes(capi.KIND_F16X2, 20, n, 600) == 0
    ctx.set_query(q_long)
    need = ctx.scan_temp_bytes(capi.KIND_F16X2, 20, n, 600)
    assert need > 0
    args = (capi.KIND_F16X2, 20, db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(), 0, n, 600, -11, -1,
            scores.data_ptr(), ids.data_ptr(), 0, ovf_pos.data_ptr(), ovf_cnt.data_ptr(), 1)
    with pytest.raises(capi.SwError) as ei:
        ctx.scan_partition(*args, 0, 0, 0)
    assert ei.value.code == -5
    expect = O.scan(q_long, chars, offsets, lengths, simd=True)
    for frac in (1.0, 0.1):
        temp = torch.empty(int(need * frac), dtype=torch.uint8, device="cuda")
        scores.fill_(-1.0)
        ctx.scan_partition(*args, temp.data_ptr(), temp.numel(), 0)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(scores.cpu().numpy().astype(np.int32), expect)
        assert ids.cpu().numpy().tolist() == list(range(n))


def test_debug_check_of_the_max_subject_len_contract(monkeypatch):
    """include/cudasw4_amd.h: an under-reported max_subject_len silently truncates; with CUDASW4_AMD_CHECK_BOUNDS=1 the
    library finds the longest subject of the range (and of an overflow list) on the device and refuses the call."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(3)
    seqs = [rng.integers(0, 20, int(l)).astype(np.int8) for l in (30, 90, 200, 301)]
    chars, offsets, lengths = O.make_db(seqs)
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    q = rng.integers(0, 20, 900).astype(np.int8)
    n = len(seqs)
    scores = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
    ids = torch.zeros(n, dtype=torch.int32, device="cuda")
    temp = torch.empty(1 << 22, dtype=torch.uint8, device="cuda")
    pos = torch.tensor([3, 0], dtype=torch.int32, device="cuda")
    cnt = torch.tensor([2], dtype=torch.int32, device="cuda")
    expect = O.scan(q, chars, offsets, lengths)
    monkeypatch.setenv("CUDASW4_AMD_CHECK_BOUNDS", "1")
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    ctx.set_query(q)

    def scan(first, cnt_n, maxlen):
        ctx.scan_partition(capi.KIND_F32, 5, db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(), first, cnt_n, maxlen,
                           -11, -1, scores.data_ptr(), ids.data_ptr(), 0, 0, 0, 0, temp.data_ptr(), temp.numel(), 0)

    def rescore(maxlen):
        ctx.rescore_overflow(capi.KIND_F32, pos.data_ptr(), cnt.data_ptr(), 2, db.chars.data_ptr(), db.offsets.data_ptr(),
                             db.lengths.data_ptr(), maxlen, -11, -1, scores.data_ptr(), ids.data_ptr(), 0, temp.data_ptr(), temp.numel(), 0)

    for call in (lambda: scan(0, n, 300), lambda: scan(2, 2, 256), lambda: rescore(300)):
        with pytest.raises(capi.SwError) as ei:
            call()
        assert ei.value.code == -1 and "under-reports" in str(ei.value) and "301" in str(ei.value)
    scan(0, 2, 90)       # the bound of THIS range, not of the DB
    scan(0, n, 301)
    rescore(301)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(scores.cpu().numpy().astype(np.int32), expect)
    ctx.close()


def test_group_shapes_agree_on_long_subjects():
    """Partitions 34/35 run with wave-wide (64-lane) groups when they hold few subjects and with 16-lane
    groups when they hold many: both shapes must give the oracle's scores."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(8)
    seqs = [rng.integers(0, 21, int(l)).astype(np.int8) for l in np.sort(rng.integers(1281, 2600, 150))]
    chars, offsets, lengths = O.make_db(seqs)
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    n = len(seqs)
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    for qlen in (130, 1100, 2100):
        q = rng.integers(0, 20, qlen).astype(np.int8)
        expect = O.scan(q, chars, offsets, lengths, simd=True)
        ctx.set_query(q)
        for kind in (capi.KIND_F16X2, capi.KIND_I16X2, capi.KIND_I32, capi.KIND_F32):
            for part_id in (33, 34):  # 33 -> 16-lane shape, 34 with a small n -> 64-lane shape
                scores = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
                ids = torch.zeros(n, dtype=torch.int32, device="cuda")
                ovf_pos = torch.zeros(n, dtype=torch.int32, device="cuda")
                ovf_cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
                need = ctx.scan_temp_bytes(kind, part_id, n, 2600)
                temp = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
                ctx.scan_partition(kind, part_id, db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(), 0, n,
                                   2600, -11, -1, scores.data_ptr(), ids.data_ptr(), 0, ovf_pos.data_ptr(),
                                   ovf_cnt.data_ptr(), 1, temp.data_ptr(), temp.numel(), 0)
                torch.cuda.synchronize()
                assert int(ovf_cnt.item()) == 0
                np.testing.assert_array_equal(scores.cpu().numpy().astype(np.int32), expect,
                                              err_msg="kind %d part %d qlen %d" % (kind, part_id, qlen))


def test_scores_beyond_int16_range_and_long_query():
    """A 9014-residue query against itself (score 46662 > 32767) and against its parts: the packed kinds must
    flag the overflow before their 16-bit state wraps and the 32-bit re-score must deliver the exact score.
    Also covers a query of 18 stripes (packed) / 24 stripes (32-bit kinds) and unknown letters in the query."""
    torch, capi, search = gpu_modules()
    _, qs = O.load_queries()
    big = np.concatenate([qs[i] for i in (10, 11, 12, 13)])
    big_x = big.copy()
    big_x[::97] = 20  # unknown residues in the query
    db = O.make_db([qs[2], qs[10], qs[13], big])
    for q in (big, big_x):
        expect = O.scan(q, *db, simd=True)
        assert expect.max() > 32767 or q is big_x
        for cfg, kt in kinds_configs(search, capi).items():
            got, res, _ = scan_all_scores(search, capi, db, q, kernel_types=kt)
            np.testing.assert_array_equal(got, expect, err_msg=cfg)
    assert O.scan(big, *db, simd=True)[3] == 46662


@pytest.mark.parametrize("gop,gex", [(-40, -10), (-1, -1), (0, 0), (-20, 0), (-2, -5)])
def test_unusual_gap_scores(gop, gex):
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(21)
    seqs = [rng.integers(0, 20, int(l)).astype(np.int8) for l in np.sort(rng.integers(5, 700, 300))]
    q = rng.integers(0, 20, 333).astype(np.int8)
    seqs[10] = q[50:250].copy()
    seqs.sort(key=len)
    db = O.make_db(seqs)
    expect = O.scan(q, *db, gop=gop, gex=gex)
    for cfg, kt in kinds_configs(search, capi).items():
        got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt, gop=gop, gex=gex)
        np.testing.assert_array_equal(got, expect, err_msg="%s gop %d gex %d" % (cfg, gop, gex))


def test_zero_length_subjects_and_single_subject_batches():
    """Empty records (length 0) score 0 and do not disturb their neighbours; counts that leave groups empty."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(31)
    for n in (1, 2, 3, 17, 33):
        seqs = [np.zeros(0, dtype=np.int8)] * min(2, n - 1) + [rng.integers(0, 20, int(l)).astype(np.int8)
                                                              for l in np.sort(rng.integers(1, 90, n - min(2, n - 1)))]
        db = O.make_db(seqs)
        q = rng.integers(0, 20, 75).astype(np.int8)
        expect = O.scan(q, *db)
        for cfg, kt in kinds_configs(search, capi).items():
            got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt)
            np.testing.assert_array_equal(got, expect, err_msg="%s n=%d" % (cfg, n))


def test_very_long_query():
    """A 70 000-residue query (137 packed stripes) against a handful of subjects, all kinds."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(41)
    q = rng.integers(0, 20, 70000).astype(np.int8)
    seqs = [rng.integers(0, 20, 40).astype(np.int8), q[30000:30300].copy(), rng.integers(0, 20, 900).astype(np.int8),
            q[100:1700].copy()]
    seqs.sort(key=len)
    db = O.make_db(seqs)
    expect = O.scan(q, *db)
    for cfg, kt in kinds_configs(search, capi).items():
        got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt)
        np.testing.assert_array_equal(got, expect, err_msg=cfg)


def test_every_compiled_tile_shape(monkeypatch):
    """Every (kind, rows-per-lane R, group shape, single/multi stripe) instantiation that the planner can pick:
    query lengths that land on each R for one and for several stripes, 16-lane groups (short subjects,
    partition 33), 64-lane groups (long subjects, partition 34 with a small n) and 8-lane groups (short queries;
    forced on / off through the environment so that both shapes see every single-stripe R).  The planner's preference
    for three-wave stripes (round 4) is switched off here, so that the tall two-wave multi-stripe kernels are reached
    too: the plans asserted below are those of the plain cost model."""
    torch, capi, search = gpu_modules()
    monkeypatch.setenv("CUDASW4_AMD_TWO_WAVE_PENALTY", "1")
    rng = np.random.default_rng(77)
    short = [rng.integers(0, 21, int(l)).astype(np.int8) for l in np.sort(rng.integers(1, 260, 37))]
    long_ = [rng.integers(0, 21, int(l)).astype(np.int8) for l in np.sort(rng.integers(1281, 1700, 9))]
    seen = set()
    for seqs, part_id, lanes in ((short, 33, 16), (long_, 34, 64), (short, 33, 8), (short, 33, 4)):
        monkeypatch.setenv("CUDASW4_AMD_I32_NATIVE", "1")  # the int32 kernels themselves, not their fp32 stand-ins
        monkeypatch.setenv("CUDASW4_AMD_LANES4_MAX_Q", "1000000" if lanes == 4 else "0")
        monkeypatch.setenv("CUDASW4_AMD_LANES8_MAX_Q", "1000000" if lanes == 8 else "0")
        ctx = capi.Context(0)  # reads the environment
        ctx.set_matrix(O.blosum21(62))
        chars, offsets, lengths = O.make_db(seqs)
        db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
        n = len(seqs)
        maxlen = int(lengths.max())
        scores = torch.empty(n, dtype=torch.float32, device="cuda")
        ids = torch.empty(n, dtype=torch.int32, device="cuda")
        ovf_pos = torch.zeros(n, dtype=torch.int32, device="cuda")
        ovf_cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        for kind in (capi.KIND_F16X2, capi.KIND_I16X2, capi.KIND_I32, capi.KIND_F32):
            # (short groups — 8 and 4 lanes — are single-stripe shapes of up to 32 rows per lane since round 6)
            rmax = {4: 32, 8: 32, 16: 48 if kind < 3 else 36, 64: 16 if kind < 2 else 8}[lanes]  # int32 stripes are as tall as the packed kinds'
            rmax_multi = 32 if (kind == 3 and lanes <= 16) else rmax  # fp32: several stripes only up to 32 rows per lane
            qlens = set()
            for r in range(1, rmax + 1):
                qlens.add(lanes * r - 1)                       # one stripe of R rows
                if lanes == 16:
                    assert capi.plan_query(kind, lanes * r - 1) == (r, 1)
                if 2 * r > rmax and r <= rmax_multi and lanes >= 16:
                    qlens.add(2 * lanes * r - lanes - 3)       # two stripes of R rows
                    if lanes == 16:
                        assert capi.plan_query(kind, 2 * lanes * r - lanes - 3) == (r, 2)
            qlens.add(lanes * rmax)                            # the longest single-stripe query of the shape
            if lanes >= 16:
                qlens.add(3 * lanes * rmax_multi - 5)          # three full stripes
            for qlen in sorted(qlens):
                q = rng.integers(0, 20, qlen).astype(np.int8)
                expect = O.scan(q, chars, offsets, lengths, simd=True)
                ctx.set_query(q)
                if lanes != 64:
                    assert ctx.plan_launch(kind, part_id, n, maxlen)[3] == lanes, (kind, lanes, qlen)
                need = ctx.scan_temp_bytes(kind, part_id, n, maxlen)
                temp = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
                scores.fill_(-1.0)
                ovf_cnt.zero_()
                ctx.scan_partition(kind, part_id, db.chars.data_ptr(), db.offsets.data_ptr(), db.lengths.data_ptr(), 0, n,
                                   maxlen, -11, -1, scores.data_ptr(), ids.data_ptr(), 0, ovf_pos.data_ptr(),
                                   ovf_cnt.data_ptr(), 1, temp.data_ptr(), temp.numel(), 0)
                torch.cuda.synchronize()
                assert int(ovf_cnt.item()) == 0
                np.testing.assert_array_equal(scores.cpu().numpy().astype(np.int32), expect,
                                              err_msg="kind %d lanes %d qlen %d" % (kind, lanes, qlen))
                seen.add((kind, lanes, qlen))
    assert len(seen) > 500


@pytest.mark.parametrize("gop,gex", [(-12, -5), (-1000, -1000), (-3, -12)])
def test_frame_lowering_on_long_subjects(gop, gex):
    """The column-offset kernels keep column j's values raised by |gex|*(j mod K + lanes) and lower the frame every K
    columns, K chosen from |gex| and the kind's range (fp16: K = 1024 at gex = -1, 128 at -5; int16: 2048 at -5; the
    32-bit kinds: 4096 at -1000).  Long subjects in both group shapes, multi- and single-stripe queries, homologs with
    large scores: every configuration must reproduce the oracle."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(99)
    lens = np.sort(np.concatenate([rng.integers(50, 1280, 40), rng.integers(1281, 7000, 30), [8100, 9300]]))
    seqs = [rng.integers(0, 20, int(l)).astype(np.int8) for l in lens]
    for qlen in (300, 1900):
        q = rng.integers(0, 20, qlen).astype(np.int8)
        for k in (5, 33, 50, 66, 70):  # homologs: the query (or a mutated copy) embedded in a long subject
            emb = q.copy()
            emb[:: 7 + k % 5] = rng.integers(0, 20, len(emb[:: 7 + k % 5]))
            pos = int(rng.integers(0, max(1, len(seqs[k]) - qlen)))
            seqs[k] = np.concatenate([seqs[k][:pos], emb, seqs[k][pos + qlen:]])[: max(len(seqs[k]), qlen)]
        db = O.make_db(seqs)
        expect = O.scan(q, *db, gop=gop, gex=gex)
        for cfg, kt in kinds_configs(search, capi).items():
            got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt, gop=gop, gex=gex)
            np.testing.assert_array_equal(got, expect, err_msg="%s gop %d gex %d qlen %d" % (cfg, gop, gex, qlen))


@pytest.mark.parametrize("qlen", [300, 600])
def test_nonfinite_packed_states(qlen):
    """With a custom matrix of large entries a subject can drive the packed state to fp16 +inf (score > 65504) or to
    an int16 NaN bit pattern (> 30719): it must come out flagged and be re-scored exactly, and its neighbours in the
    wave must not be affected."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(5)
    m = np.full((21, 21), -4, dtype=np.int8)
    np.fill_diagonal(m, 120)
    m[20, 20] = -4                              # scores against the padding letter must stay negative
    q = rng.integers(0, 20, qlen).astype(np.int8)
    seqs = [rng.integers(0, 20, int(l)).astype(np.int8) for l in rng.integers(20, qlen + 40, 700)]
    for k in range(0, 700, 37):
        seqs[k] = q.copy()                      # self hits: qlen * 120 = 36000 / 72000
    for k in range(5, 700, 53):
        seqs[k] = q[: qlen // 2].copy()         # half hits: 18000 / 36000
    db = O.make_db(seqs)
    expect = O.scan(q, *db, m21=m, simd=False)
    assert expect.max() == qlen * 120
    for cfg, kt in kinds_configs(search, capi).items():
        got, res, _ = scan_all_scores(search, capi, db, q, kernel_types=kt, matrix=m)
        np.testing.assert_array_equal(got, expect, err_msg=cfg)


def test_randomized_stress_many_scans():
    """Back-to-back scans with changing query lengths and kernel configurations on one resident ragged DB
    (concurrent partition launches, profile rebuilds, work-counter reuse, overflow re-score): every scan must
    reproduce the oracle, including the overflow count and the top-K."""
    torch, capi, search = gpu_modules()
    rng = np.random.default_rng(1234)
    lens = np.sort(np.concatenate([rng.integers(1, 400, 1500), rng.integers(400, 1280, 300), rng.integers(1281, 4000, 40),
                                   [8200, 9100]]))
    seqs = [rng.integers(0, 20, int(l)).astype(np.int8) for l in lens]
    chars, offsets, lengths = O.make_db(seqs)
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    cfgs = list(kinds_configs(search, capi).items())
    searchers = {}
    for name, kt in cfgs:
        s = search.Searcher(device=0, num_top=15, matrix=O.blosum21(62), kernel_types=kt)
        s.set_database(db)
        searchers[name] = s
    for it in range(60):
        qlen = int(rng.choice([rng.integers(1, 64), rng.integers(64, 600), rng.integers(600, 2500)]))
        q = rng.integers(0, 20, qlen).astype(np.int8)
        if it % 3 == 0:  # a homolog of a long DB entry: large scores, packed overflows
            src = seqs[int(rng.integers(len(seqs) - 45, len(seqs)))]
            q = src[: max(1, min(len(src), qlen))].copy()
        expect = O.scan(q, chars, offsets, lengths, simd=True)
        es, ei = O.topk(expect, 15)
        name, _ = cfgs[it % len(cfgs)]
        s = searchers[name]
        res = s.scan(q)
        np.testing.assert_array_equal(s.all_scores(), expect, err_msg="%s it %d qlen %d" % (name, it, len(q)))
        assert res.scores.tolist() == es.tolist() and res.reference_ids.tolist() == ei.tolist(), (name, it)
        packed = lengths <= 8000  # partition 35 (> 8000) is scored by a 32-bit kind in these configurations
        if name == "half2+float":
            check_overflow_count(res.num_overflows, expect[packed], lengths[packed], 2048)
        if name == "dpxs16+dpxs32":
            check_overflow_count(res.num_overflows, expect[packed], lengths[packed], 25000)


def full_matrix_as_oracle_rows(m25):
    """The 25 x 25 table as the oracle takes it: one row per QUERY letter (25), 21 columns = the dbdata alphabet of the
    subjects, where code 20 ("other") is scored with the table's X column (include/cudasw4_amd.h, sw_set_matrix)."""
    m = np.asarray(m25, dtype=np.int8).reshape(25, 25)
    cols = list(range(20)) + [23]
    return np.ascontiguousarray(m[:, cols])


@pytest.mark.parametrize("which", [62, 45])
def test_full_25_letter_matrix_all_kinds(which):
    """SURVEY §8 f4 / the reference's CAN_USE_FULL_BLOSUM build (options.cpp:135-143, types.hpp:205-396): a 25 x 25
    table through sw_set_matrix; queries with all 25 letters (B, J, Z, X, * score differently), subjects in the dbdata
    alphabet incl. code 20; single- and multi-stripe queries, both group shapes, all four kinds, vs the scalar oracle."""
    torch, capi, search = gpu_modules()
    from cudasw4_amd import driver
    rng = np.random.default_rng(1234 + which)
    m25 = driver.matrix25(which)
    mo = full_matrix_as_oracle_rows(m25)
    lens = np.sort(np.concatenate([rng.integers(1, 600, 300), rng.integers(1281, 3000, 12), [8200]]))
    seqs = [rng.integers(0, 21, int(l)).astype(np.int8) for l in lens]
    db = O.make_db(seqs)
    gop, gex = (-11, -1) if which == 62 else (-13, -2)
    for qlen in (37, 300, 769, 1700):
        q = rng.integers(0, 25, qlen).astype(np.int8)
        q[::7] = 24  # plenty of '*' (the +1 diagonal entry) and of the ambiguity letters
        q[3::11] = 20
        expect = O.scan(q, *db, m21=mo, gop=gop, gex=gex)
        for cfg, kt in kinds_configs(search, capi).items():
            got, _, _ = scan_all_scores(search, capi, db, q, kernel_types=kt, matrix=m25, gop=gop, gex=gex)
            np.testing.assert_array_equal(got, expect, err_msg="%s qlen %d" % (cfg, qlen))
    # a query of the 21-letter alphabet scores the same under both tables whenever the subject has no code 20
    clean = [s for s in seqs[:200] if (s < 20).all() and len(s) > 0][:50]
    dbc = O.make_db(clean)
    q = rng.integers(0, 20, 400).astype(np.int8)
    a, _, _ = scan_all_scores(search, capi, dbc, q, matrix=m25, gop=gop, gex=gex)
    b, _, _ = scan_all_scores(search, capi, dbc, q, matrix=driver.matrix(which), gop=gop, gex=gex)
    np.testing.assert_array_equal(a, b)
    # codes outside the installed alphabet are refused
    ctx = capi.Context(0)
    ctx.set_matrix(driver.matrix(which))
    with pytest.raises(capi.SwError):
        ctx.set_query(np.array([0, 22, 3], dtype=np.int8))
    ctx.set_matrix(m25)
    ctx.set_query(np.array([0, 22, 3], dtype=np.int8))
    with pytest.raises(capi.SwError):
        ctx.set_query(np.array([0, 25], dtype=np.int8))


@pytest.mark.parametrize("native", ["0", "1"])
def test_int32_kind_native_and_in_fp32_lanes(native, monkeypatch):
    """The int32 kind is served by the fp32 kernels whenever min(query, subject) * max(matrix) + 2^22 < 2^24 proves the
    fp32 arithmetic exact (sw_api.hip: effective_kind); CUDASW4_AMD_I32_NATIVE=1 keeps the int32 kernels.  Both must give
    the oracle's scores: ragged DB with long subjects, homologs scoring far above the 16-bit ranges, all partitions int32."""
    torch, capi, search = gpu_modules()
    monkeypatch.setenv("CUDASW4_AMD_I32_NATIVE", native)
    rng = np.random.default_rng(4242)
    _, qs = O.load_queries()
    lens = np.sort(np.concatenate([rng.integers(1, 1280, 300), rng.integers(1281, 6000, 20), [9000]]))
    seqs = [rng.integers(0, 21, int(l)).astype(np.int8) for l in lens]
    seqs[310][:len(qs[17])] = qs[17][:len(seqs[310])]   # a homolog of query 17 (4743 residues): score in the tens of thousands
    db = O.make_db(seqs)
    kt = search.KernelTypeConfig(capi.KIND_I32, capi.KIND_I16X2, capi.KIND_I32, capi.KIND_I32)
    for qi in (0, 6, 17):
        expect = O.scan(qs[qi], *db, simd=True)
        got, res, _ = scan_all_scores(search, capi, db, qs[qi], kernel_types=kt)
        np.testing.assert_array_equal(got, expect, err_msg="native %s query %d" % (native, qi))
    assert expect.max() > 10000


def test_letter_code_guard_and_launch_plan_introspection(monkeypatch):
    """The two auxiliary entry points of the C ABI.  sw_check_letter_codes: flags a byte outside 0..20 wherever it sits
    (aligned body, ragged head and tail, negative bytes), passes clean arrays of any size and alignment.
    sw_plan_launch: names the instantiation the next scan would launch — stripes x rows x lanes cover the query, short
    queries get 8-lane groups, the long partitions the wave-wide shape when they hold few subjects, and an int32 request
    is served in fp32 lanes unless the score bound (or CUDASW4_AMD_I32_NATIVE) forbids it."""
    torch, capi, search = gpu_modules()
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    rng = np.random.default_rng(3)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for n in (1, 3, 15, 16, 17, 31, 64, 100, 4097, 1 << 20):
        for shift in (0, 1, 5, 15):
            host = rng.integers(0, 21, n + shift).astype(np.int8)
            dev = torch.from_numpy(host).cuda()
            flag.zero_()
            ctx.check_letter_codes(dev.data_ptr() + shift, n, flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert int(flag.item()) == 0, (n, shift)
            for pos, bad in ((0, 21), (n - 1, 127), (n // 2, -1), (min(n - 1, 17), -128), (n - 1, 32)):
                h2 = host.copy()
                h2[shift + pos] = bad
                d2 = torch.from_numpy(h2).cuda()
                flag.zero_()
                ctx.check_letter_codes(d2.data_ptr() + shift, n, flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert int(flag.item()) == 1, (n, shift, pos, bad)
    ctx.check_letter_codes(0, 0, flag.data_ptr())   # nothing to check

    def plan(qlen, kind, part, n, maxlen):
        ctx.set_query(rng.integers(0, 20, qlen).astype(np.int8), torch.cuda.current_stream().cuda_stream)
        return ctx.plan_launch(kind, part, n, maxlen)

    for qlen in (1, 48, 144, 256, 257, 375, 768, 769, 850, 5478, 20000):
        for kind in (capi.KIND_F16X2, capi.KIND_I16X2, capi.KIND_I32, capi.KIND_F32):
            k, rows, ns, lanes = plan(qlen, kind, 20, 100000, 512)
            assert rows * lanes * ns >= qlen and rows >= 1 and ns >= 1 and lanes in (8, 16)
            assert k == (capi.KIND_F32 if kind == capi.KIND_I32 else kind)           # 512 x 11 + 2^22 < 2^24
            packed = kind in (capi.KIND_F16X2, capi.KIND_I16X2)
            assert lanes == (8 if qlen <= 256 else 16), (qlen, kind, lanes)
            assert (rows, ns) == capi.plan_query(k, qlen) or lanes == 8
    assert plan(5478, capi.KIND_F32, 35, 19, 35213)[3] == 64          # a few giants: wave-wide groups
    assert plan(5478, capi.KIND_F16X2, 34, 400, 8000)[3] == 64 and plan(5478, capi.KIND_F16X2, 34, 600, 8000)[3] == 16
    assert plan(5478, capi.KIND_F32, -1, 1000, 700)[3] == 16 and plan(5478, capi.KIND_F32, -1, 1000, 1500)[3] == 64   # re-score
    assert plan(20000, capi.KIND_I32, 35, 10, 2_000_000)[0] == capi.KIND_F32      # 20000 x 11 + 2^22 < 2^24: exact in fp32
    assert plan(2_000_000, capi.KIND_I32, 35, 10, 2_000_000)[0] == capi.KIND_I32  # 2 * 10^6 x 11 is not: the int32 kernels
    assert plan(1000, capi.KIND_I32, 35, 10, 2_000_000)[0] == capi.KIND_F32
    ctx.close()
    monkeypatch.setenv("CUDASW4_AMD_I32_NATIVE", "1")
    ctx = capi.Context(0)
    ctx.set_matrix(O.blosum21(62))
    ctx.set_query(rng.integers(0, 20, 300).astype(np.int8), torch.cuda.current_stream().cuda_stream)
    assert ctx.plan_launch(capi.KIND_I32, 20, 1000, 512)[0] == capi.KIND_I32
    ctx.close()
