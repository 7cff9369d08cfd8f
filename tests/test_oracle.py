"""CPU: the oracle restatement against the golden vectors produced by the reference itself."""
import os

import numpy as np
import pytest

import oracle_lib as O


def test_tables_match_reference():
    g = O.golden("ref_tables.json")
    for which in (45, 50, 62, 80):
        assert O.blosum21(which).tolist() == g["blosum21"][str(which)]
    assert O.partition_boundaries().tolist() == g["partition_boundaries"]
    assert len(g["partition_boundaries"]) == 36


def test_encoder_matches_reference():
    g = O.golden("ref_tables.json")
    got = O.encode(bytes(range(256)))
    assert got.tolist() == g["encode_map_256"]
    assert O.encode("ARNDCQEGHILKMFPSTWYV").tolist() == list(range(20))
    assert O.encode("XBZ*a- ").tolist() == [20] * 7


def test_pseudodb_generator_matches_reference():
    g = O.golden("ref_tables.json")
    assert O.pseudodb_codes(2048, 42).tolist() == g["pseudodb_seed42_first2048"]
    assert O.pseudodb_codes(64, 7).tolist() == g["pseudodb_seed7_first64"]
    # every length shares the same stream prefix (dbdata.hpp:233-240)
    assert O.pseudodb_codes(128, 42).tolist() == g["pseudodb_seed42_first2048"][:128]


def test_queries_parse_like_reference():
    g = O.golden("ref_tables.json")
    headers, qs = O.load_queries()
    assert [len(q) for q in qs] == g["query_lengths"]
    assert headers == g["query_headers"]
    assert sum(len(q) for q in qs) == 41752


def test_score_pairs_match_reference_dp():
    g = O.golden("ref_scores.json")
    for p in g["pairs"]:
        got = O.score(np.array(p["q"], dtype=np.int8), np.array(p["s"], dtype=np.int8),
                      gop=p.get("gop", -11), gex=p.get("gex", -1))
        assert got == p["score"], (len(p["q"]), len(p["s"]))


def test_pseudo_scores_match_reference_dp():
    g = O.golden("ref_scores.json")
    _, qs = O.load_queries()
    for L, expect in g["pseudo"].items():
        subj = O.pseudodb_codes(int(L), 42)
        assert [O.score(q, subj) for q in qs] == expect


def test_allvsall_matches_reference_dp_scalar_and_simd():
    g = O.golden("ref_scores.json")
    _, qs = O.load_queries()
    chars, offsets, lengths = O.make_db(qs)
    expect = np.array(g["allvsall"], dtype=np.int32)
    for i in (0, 3, 4, 9, 16, 19):       # includes scores >= 2048 and >= 25000
        assert O.scan(qs[i], chars, offsets, lengths).tolist() == expect[i].tolist()
    for i in range(len(qs)):              # SIMD baseline incl. int16 saturation re-score
        assert O.scan(qs[i], chars, offsets, lengths, simd=True).tolist() == expect[i].tolist()


def test_long_subject_matches_reference_dp():
    g = O.golden("ref_scores.json")["long_subject"]
    _, qs = O.load_queries()
    subj = np.concatenate([qs[i] for i in g["concat_of_queries"]])
    assert len(subj) == g["length"] > 8000
    for i in (0, 5, 10, 13, 19):
        assert O.score(qs[i], subj) == g["scores"][i]


def test_score_is_symmetric_and_handles_empty():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 21, 77).astype(np.int8)
    b = rng.integers(0, 21, 130).astype(np.int8)
    assert O.score(a, b) == O.score(b, a)
    assert O.score(a, b[:0]) == 0
    assert O.score(a[:0], b) == 0


def test_simd_equals_scalar_on_ragged_db():
    rng = np.random.default_rng(7)
    lens = np.sort(rng.integers(1, 400, 203))
    seqs = [rng.integers(0, 21, int(n)).astype(np.int8) for n in lens]
    chars, offsets, lengths = O.make_db(seqs)
    q = rng.integers(0, 20, 150).astype(np.int8)
    np.testing.assert_array_equal(O.scan(q, chars, offsets, lengths), O.scan(q, chars, offsets, lengths, simd=True))


def test_striped_equals_scalar():
    """Farrar striped SW (second CPU baseline): lazy-F path, query lengths around the vector width, gap
    settings where extension costs more than opening, scores that saturate int16."""
    rng = np.random.default_rng(11)
    lens = np.sort(rng.integers(1, 400, 150))
    seqs = [rng.integers(0, 21, int(n)).astype(np.int8) for n in lens]
    seqs[100] = seqs[140][: len(seqs[100])].copy()
    db = O.make_db(seqs)
    for ql in (1, 15, 16, 17, 31, 32, 33, 150, 513):
        q = rng.integers(0, 20, ql).astype(np.int8)
        if ql == 150:
            q = seqs[140][:150].copy()
        for gop, gex in ((-11, -1), (-3, -3), (0, 0), (-1, -5), (-20, -7)):
            np.testing.assert_array_equal(O.scan(q, *db, gop=gop, gex=gex), O.scan(q, *db, gop=gop, gex=gex, striped=True),
                                          err_msg="q %d gop %d gex %d" % (ql, gop, gex))
    g = O.golden("ref_scores.json")
    _, qs = O.load_queries()
    dbq = O.make_db(qs)
    for i in (3, 16, 19):  # incl. scores >= 25000 -> int32 re-score
        assert O.scan(qs[i], *dbq, striped=True).tolist() == g["allvsall"][i]


def test_topk_order():
    s = np.array([5, 9, 9, 1, 7], dtype=np.int32)
    sc, ids = O.topk(s, 3)
    assert sc.tolist() == [9, 9, 7] and ids.tolist() == [1, 2, 4]
    sc, ids = O.topk(s, 7)
    assert sc.tolist()[5:] == [-1, -1]


def test_partition_of():
    b = O.partition_boundaries()
    assert O.lib().swo_partition_of(1) == 0
    assert O.lib().swo_partition_of(48) == 0
    assert O.lib().swo_partition_of(49) == 1
    assert O.lib().swo_partition_of(1280) == 33
    assert O.lib().swo_partition_of(1281) == 34
    assert O.lib().swo_partition_of(8000) == 34
    assert O.lib().swo_partition_of(8001) == 35
    assert b[35] == 2**31 - 2


@pytest.mark.skipif(not os.path.exists(os.path.join(O.ORACLE_DIR, "_ref", "libref_dp.so")),
                    reason="oracle/_ref not built (reference tree absent)")
def test_oracle_equals_reference_dp_live_random():
    """When oracle/_ref is present: fuzz the restatement against the reference's own DP."""
    import ctypes
    dp = ctypes.CDLL(os.path.join(O.ORACLE_DIR, "_ref", "libref_dp.so"))
    rng = np.random.default_rng(99)
    for _ in range(200):
        a = rng.integers(0, 21, int(rng.integers(1, 200))).astype(np.int8)
        b = rng.integers(0, 21, int(rng.integers(1, 200))).astype(np.int8)
        ref = dp.ref_dp_score_converted(a.tobytes(), b.tobytes(), len(a), len(b), -11, -1)
        assert O.score(a, b) == ref
