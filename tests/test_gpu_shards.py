"""GPU: the shapes BASELINE.json configs 4 and 5 take on an 8-GPU node, exercised on this box's one GPU.

config 4  one DB cut into 8 char-balanced shards per length partition (partitionDBAmongstGpus, cudasw4.cuh:928-1004),
          one host worker thread and one context per shard, per-shard top-K, host merge (cudasw4.cuh:1415-1463)
config 5  the same with shards above the memory limit: hybrid residency (part of every shard cached in device memory,
          the rest streamed per query, cudasw4.cuh:1044-1046,1087-1144,1565-1621)
plus the two-halves scan (submit / collect) that keeps the GPUs busy across query boundaries.
Everything goes through the C++ host driver -> C ABI -> HIP kernels and is compared with the CPU oracle.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

FASTA = os.path.join(O.GOLDEN_DIR, "allqueries.fasta")


def sample_db(chars, offsets, lengths, pick):
    from cudasw4_amd import search
    return search.build_shard(chars, offsets, lengths, [(int(i), int(i) + 1) for i in pick])[:3]


@pytest.fixture(scope="module")
def sprot_db():
    from cudasw4_amd import synthdb
    return synthdb.sprot_like()  # 570 000 sequences, ~2e8 residues, seeded


def check_eight_shards(d, n, total_residues):
    infos = [d.shard_info(g) for g in range(8)]
    assert d.num_gpus() == 8 and sum(i["subjects"] for i in infos) == n
    res = np.array([i["residues"] for i in infos], dtype=np.float64)
    assert res.sum() == total_residues
    # char-balanced per length partition: the shards' work differs by a few percent at most
    assert res.max() / res.min() < 1.05, res
    return infos


def spans_overlap(d):
    spans = d.gpu_spans()
    assert len(spans) == 8
    assert max(b for b, e in spans) < min(e for b, e in spans), spans


@pytest.mark.parametrize("mode", ["resident", "hybrid", "streamed"])
def test_eight_shards_sprot_like(sprot_db, mode, monkeypatch):
    """Config 4's shape on the 570 k-sequence Swiss-Prot-like DB: 8 shards of one GPU, half2 / float kernels.  Every
    score of a seeded sample that includes the whole > 8000 tail (16 giants: most shards get none or one of them)
    equals the oracle, the merged top-25 equals the top of ALL scores and its winners are oracle-checked, the shards
    are balanced and run concurrently.  hybrid / streamed: the same with the shards above the memory limit."""
    from cudasw4_amd import driver, search
    chars, offsets, lengths = sprot_db
    n = len(lengths)
    _, letters = O.read_fasta(FASTA)
    kw = {}
    if mode == "hybrid":
        kw = dict(max_gpu_mem=30 << 20, max_batch_bytes=2 << 20)   # a shard is ~27 MB of chars
    elif mode == "streamed":
        kw = dict(max_gpu_mem=1, max_batch_bytes=4 << 20)
    d = driver.Driver(devices=[0] * 8, num_top=25, kinds=(0, 0, 3, 3), **kw)
    d.db_from_arrays(chars, offsets, lengths)
    d.upload()
    infos = check_eight_shards(d, n, int(lengths.astype(np.int64).sum()))
    if mode == "resident":
        assert all(i["resident"] and i["cached_chars"] == i["chars"] for i in infos)
    elif mode == "hybrid":
        assert all(not i["resident"] and 0.3 * i["chars"] < i["cached_chars"] < 0.9 * i["chars"] for i in infos), infos
    else:
        assert all(not i["resident"] and i["cached_chars"] == 0 for i in infos)
    rng = np.random.default_rng(17)
    tail = np.nonzero(lengths > 8000)[0]
    assert 8 <= len(tail) <= 64
    pick = np.unique(np.concatenate([rng.choice(n, 1500, replace=False), tail]))
    sub = sample_db(chars, offsets, lengths, pick)
    before = d.streamed_bytes()
    for qi in (0, 9, 19):
        r = d.scan(letters[qi])
        ids, sc = d.all_scores()
        full = np.empty(n, dtype=np.int32)
        full[ids] = sc
        expect = O.scan(O.encode(letters[qi]), *sub, simd=True)
        assert (full[pick] == expect).all(), (mode, qi, np.nonzero(full[pick] != expect)[0][:5])
        es, ei = search.merge_topk([(full, np.arange(n))], 25)
        assert r["scores"].tolist() == es.tolist() and r["ids"].tolist() == ei.tolist(), (mode, qi)
        top_sub = sample_db(chars, offsets, lengths, np.sort(ei))
        assert sorted(O.scan(O.encode(letters[qi]), *top_sub, simd=True).tolist(), reverse=True) == es.tolist()
        spans_overlap(d)
    moved = d.streamed_bytes() - before
    streamed_per_query = sum(i["chars"] - i["cached_chars"] for i in infos)
    if mode == "resident":
        assert moved == 0
    else:
        # every query moves the overhang once (+ at most one batch per shard staged ahead for the query after the last)
        assert 3 * streamed_per_query <= moved <= 3 * streamed_per_query + 8 * (4 << 20) + 8 * 64
    d.close()


@pytest.mark.parametrize("mode", ["resident", "hybrid"])
def test_eight_shards_pseudo_db_full_size(mode):
    """The peak benchmark's DB (10^6 x 512) in 8 shards: all 10^6 scores of a query equal the reference's golden score,
    the merged top-25 are the 25 lowest ids, 125 000 subjects per shard."""
    from cudasw4_amd import driver
    golden = O.golden("ref_scores.json")["pseudo"]["512"]
    _, letters = O.read_fasta(FASTA)
    kw = dict(max_gpu_mem=60 << 20, max_batch_bytes=8 << 20) if mode == "hybrid" else {}   # a shard is 64 MB of chars
    d = driver.Driver(devices=[0] * 8, num_top=25, kinds=(0, 0, 3, 3), **kw)
    d.pseudo_db(1_000_000, 512)
    d.upload()
    infos = check_eight_shards(d, 1_000_000, 512_000_000)
    assert all(abs(i["subjects"] - 125_000) <= 16 for i in infos), infos   # a slice ends at the first subject past its quota
    if mode == "hybrid":
        assert all(0 < i["cached_chars"] < i["chars"] for i in infos)
    for qi in (2, 12, 19):
        r = d.scan(letters[qi])
        ids, sc = d.all_scores()
        assert len(ids) == 1_000_000 and (np.sort(ids) == np.arange(1_000_000)).all()
        assert int(sc.min()) == int(sc.max()) == int(golden[qi])
        assert r["scores"].tolist() == [int(golden[qi])] * 25 and r["ids"].tolist() == list(range(25))
        spans_overlap(d)
    d.close()


def test_hybrid_residency_moves_only_the_overhang(monkeypatch):
    """§8 f3 (cudasw4.cuh:1044-1046,1087-1144,1565-1621): a shard above the memory limit keeps as much as fits in
    device memory (the longest subjects) and streams only the rest.  With the limit set for ~60 % of the shard the
    bytes copied host -> device per query — counted by the driver — are exactly the streamed remainder from the second
    query on, every score equals the oracle, and the result is the resident driver's."""
    from cudasw4_amd import driver, synthdb
    lengths = synthdb.sprot_like_lengths(60000, seed=21, max_len=15000)
    chars, offsets, lengths = synthdb.random_db(lengths, seed=22, other_fraction=0.01)
    nchars = int(offsets[-1])
    _, letters = O.read_fasta(FASTA)
    batch = 1 << 20
    # the limit covers metadata (24 B per subject), scratch (a quarter), three staging buffers and the cached chars
    want_cached = 0.6 * nchars
    limit = int((want_cached + 3 * (batch + 64) + 64) / 0.75) + 24 * len(lengths) + 8
    expect = {qi: O.scan(O.encode(letters[qi]), chars, offsets, lengths, simd=True) for qi in (1, 8, 15, 19)}
    ref = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2))
    ref.db_from_arrays(chars, offsets, lengths)
    ref.upload()
    for hostreg in ("0", "1"):
        monkeypatch.setenv("CUDASW4_AMD_NO_HOSTREGISTER", hostreg)
        d = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2), max_gpu_mem=limit, max_batch_bytes=batch)
        d.db_from_arrays(chars, offsets, lengths)
        info = d.shard_info(0)
        assert not info["resident"] and info["chars"] == nchars
        assert 0.55 * nchars <= info["cached_chars"] <= 0.62 * nchars, info
        streamed = info["chars"] - info["cached_chars"]
        prev = None
        for qi in (1, 8, 15, 19):
            before = d.streamed_bytes()
            r = d.scan(letters[qi])
            moved = d.streamed_bytes() - before
            if prev is not None:
                assert moved == streamed, (hostreg, qi, moved, streamed)   # ~40 % of the chars, not all of them
            prev = moved
            ids, sc = d.all_scores()
            assert (sc[np.argsort(ids)] == expect[qi]).all(), (hostreg, qi)
            rr = ref.scan(letters[qi])
            assert r["scores"].tolist() == rr["scores"].tolist() and r["ids"].tolist() == rr["ids"].tolist()
            assert r["num_overflows"] == rr["num_overflows"]
        d.close()
    # the switch that restores all-or-nothing residency (A/B measurements)
    monkeypatch.setenv("CUDASW4_AMD_NO_HYBRID", "1")
    d = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2), max_gpu_mem=limit, max_batch_bytes=batch)
    d.db_from_arrays(chars, offsets, lengths)
    assert d.shard_info(0)["cached_chars"] == 0
    r = d.scan(letters[8])
    ids, sc = d.all_scores()
    assert (sc[np.argsort(ids)] == expect[8]).all()
    d.close()
    ref.close()


@pytest.mark.parametrize("devices,kw", [([0], {}), ([0, 0, 0], dict(max_gpu_mem=1, max_batch_bytes=300_000)),
                                         ([0, 0], dict(max_gpu_mem=6 << 20, max_batch_bytes=400_000))])
def test_submit_collect_pipeline_equals_scan(devices, kw):
    """SearchDriver::submit / collect (two queries in flight, what `align` does with a query file): the same results
    as one scan after the other, for resident, streamed and hybrid shards; the misuse cases are refused."""
    from cudasw4_amd import driver, synthdb
    lengths = synthdb.sprot_like_lengths(20000, seed=31, max_len=10000)
    chars, offsets, lengths = synthdb.random_db(lengths, seed=32, other_fraction=0.01)
    _, letters = O.read_fasta(FASTA)
    d = driver.Driver(devices=devices, num_top=15, kinds=(0, 0, 3, 3), **kw)
    d.db_from_arrays(chars, offsets, lengths)
    one_by_one = [d.scan(q) for q in letters]
    for _ in range(2):
        piped = d.scan_many(letters)
        assert len(piped) == len(letters)
        for a, b in zip(one_by_one, piped):
            assert a["scores"].tolist() == b["scores"].tolist() and a["ids"].tolist() == b["ids"].tolist()
            assert a["num_overflows"] == b["num_overflows"] and a["num_rescored"] == b["num_rescored"]
            assert b["seconds"] > 0 and b["gcups"] > 0
    # winners of three queries against the oracle
    for qi in (0, 10, 19):
        es, ei = O.topk(O.scan(O.encode(letters[qi]), chars, offsets, lengths, simd=True), 15)
        assert piped[qi]["scores"].tolist() == es.tolist() and piped[qi]["ids"].tolist() == ei.tolist()
    with pytest.raises(driver.DriverError):
        d.collect()                       # nothing submitted
    for qi in range(4):                   # SearchDriver::kMaxInFlight = 4 (round 5) may be pending
        d.submit(letters[qi])
    with pytest.raises(driver.DriverError):
        d.submit(letters[4])              # a fifth query in flight
    with pytest.raises(driver.DriverError):
        d.scan(letters[4])                # scan() while queries are pending
    got = [d.collect() for _ in range(4)]
    for qi in range(4):
        assert got[qi]["scores"].tolist() == one_by_one[qi]["scores"].tolist() and got[qi]["ids"].tolist() == one_by_one[qi]["ids"].tolist()
    # the last scan's scores are those of the query collected last
    ids, sc = d.all_scores()
    expect = O.scan(O.encode(letters[3]), chars, offsets, lengths, simd=True)
    assert (sc[np.argsort(ids)] == expect).all()
    d.close()


@pytest.mark.parametrize("streamed", [False, True])
def test_unvalidated_corrupt_db_is_refused_on_the_device(tmp_path, monkeypatch, streamed):
    """A memory-mapped DB too large to validate at load skips the host pass over its chars (here forced with
    CUDASW4_AMD_VALIDATE_DB=0): the driver then checks the letter codes on the device as the chars arrive — the cached
    part at its upload, a streamed batch the first time it is scanned — and refuses a DB with codes outside 0..20
    instead of returning garbage scores."""
    import shutil
    from cudasw4_amd import driver
    src = os.path.join(O.GOLDEN_DIR, "allqueries_db")
    dst = tmp_path / "db"
    shutil.copytree(src, dst)
    prefix = str(dst / "aq")
    _, letters = O.read_fasta(FASTA)
    monkeypatch.setenv("CUDASW4_AMD_VALIDATE_DB", "0")
    kw = dict(max_gpu_mem=1, max_batch_bytes=6000) if streamed else {}
    good = driver.Driver(devices=[0], num_top=3, kinds=(0, 0, 3, 3), **kw)
    good.open_db(prefix)
    good.upload()
    assert good.scan(letters[3])["scores"][0] == O.golden("ref_scores.json")["allvsall"][3][3]
    good.close()
    with open(prefix + "0chars", "r+b") as f:
        f.seek(1234)
        f.write(bytes([77]))
    d = driver.Driver(devices=[0], num_top=3, kinds=(0, 0, 3, 3), **kw)
    d.open_db(prefix)      # not noticed at load: the host pass is skipped
    with pytest.raises(driver.DriverError, match="letter codes"):
        d.upload()
        d.scan(letters[3])
    with pytest.raises(driver.DriverError, match="letter codes"):
        d.scan(letters[3])  # and it stays refused
    d.close()
