"""CPU: makedb / dbdata layout parity with the reference, CLI surface, driver library exports."""
import ctypes
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = O.ROOT
LIBDIR = os.path.join(ROOT, "cudasw4_amd", "lib")
MAKEDB = os.path.join(LIBDIR, "makedb")
ALIGN = os.path.join(LIBDIR, "align")
REF_MAKEDB = os.path.join(ROOT, "oracle", "_ref", "makedb")
DB_FILES = ["0chars", "0offsets", "0lengths", "0headers", "0headeroffsets", "0metadata", "metadata"]


@pytest.fixture(scope="module", autouse=True)
def built():
    if not (os.path.exists(MAKEDB) and os.path.exists(ALIGN)):
        import __graft_entry__ as g
        g.build()


def run_makedb(exe, fasta, prefix):
    subprocess.check_call([exe, fasta, prefix], stdout=subprocess.DEVNULL)
    return {f: open(prefix + f, "rb").read() for f in DB_FILES}


def test_makedb_matches_reference_golden_files(tmp_path, golden_dir):
    """Byte-for-byte equality with the DB the reference's makedb wrote for allqueries.fasta."""
    got = run_makedb(MAKEDB, os.path.join(golden_dir, "allqueries.fasta"), str(tmp_path / "aq"))
    for f in DB_FILES:
        ref = open(os.path.join(golden_dir, "allqueries_db", "aq" + f), "rb").read()
        assert got[f] == ref, f
    lengths = np.frombuffer(got["0lengths"], dtype=np.int32)
    assert lengths.tolist() == sorted(lengths.tolist()) and len(lengths) == 20
    meta = np.frombuffer(got["0metadata"][:4 + 36 * 4], dtype=np.int32)
    assert meta[0] == 36 and meta[1:].tolist() == O.partition_boundaries().tolist()


def _random_fasta(path, rng, n, crlf=False, fastq=False, gz=False):
    letters = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVXBZ*acd", dtype=np.uint8)
    out = []
    nl = "\r\n" if crlf else "\n"
    for i in range(n):
        L = int(rng.integers(1, 90)) if i % 7 else int(rng.integers(90, 1500))
        seq = bytes(rng.choice(letters, L)).decode()
        if fastq:
            out.append("@read%d some description%s%s%s+%s%s%s" % (i, nl, seq, nl, nl, "I" * L, nl))
        else:
            wrapped = nl.join(seq[k:k + 60] for k in range(0, L, 60))
            out.append(">sp|P%05d|NAME_%d OS=Test organism%s%s%s" % (i, i, nl, wrapped, nl))
            if i % 11 == 0:
                out.append(nl)  # empty line between records
    data = "".join(out).encode()
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


@pytest.mark.skipif(not os.path.exists(REF_MAKEDB), reason="reference makedb not built (oracle/_ref)")
@pytest.mark.parametrize("variant", ["plain", "crlf", "gz", "fastq"])
def test_makedb_equals_reference_makedb_live(tmp_path, variant):
    """Same input through both binaries (length ties, unknown letters, wrapped lines, CRLF, gzip, FASTQ)."""
    rng = np.random.default_rng(11)
    fasta = str(tmp_path / ("in.fa.gz" if variant == "gz" else "in.fa"))
    _random_fasta(fasta, rng, 400, crlf=variant == "crlf", fastq=variant == "fastq", gz=variant == "gz")
    mine = run_makedb(MAKEDB, fasta, str(tmp_path / "mine"))
    ref = run_makedb(REF_MAKEDB, fasta, str(tmp_path / "ref"))
    for f in DB_FILES:
        assert mine[f] == ref[f], (variant, f)


def test_makedb_memory_limit_spills_and_matches(tmp_path):
    """--mem below the input size: the batch spills to <tempdir>/_cudasw4tmp* files (makedb.cpp:90-94) and
    the DB is byte-identical to the in-memory run; temp files are removed afterwards."""
    rng = np.random.default_rng(23)
    fasta = str(tmp_path / "in.fa")
    _random_fasta(fasta, rng, 3000)
    plain = run_makedb(MAKEDB, fasta, str(tmp_path / "plain"))
    tmpd = tmp_path / "tmp"
    tmpd.mkdir()
    out = subprocess.run([MAKEDB, fasta, str(tmp_path / "lim"), "--mem", "64K", "--tempdir", str(tmpd)],
                         capture_output=True, text=True)
    assert out.returncode == 0 and "Memory limit reached" in out.stdout and "availableMem: 65536" in out.stdout
    for f in DB_FILES:
        assert open(str(tmp_path / "lim") + f, "rb").read() == plain[f], f
    assert list(tmpd.iterdir()) == []


def test_makedb_parallel_sort_is_stable_and_equivalent(tmp_path):
    """The parallel sort (inputs of 2^20 sequences and more; forced here by the environment override) keeps equal
    lengths in input order and otherwise writes the same DB: same lengths, same multiset of (header, sequence)."""
    rng = np.random.default_rng(31)
    fasta = str(tmp_path / "in.fa")
    _random_fasta(fasta, rng, 2000)
    serial = run_makedb(MAKEDB, fasta, str(tmp_path / "ser"))
    env = dict(os.environ, CUDASW4_AMD_PARALLEL_SORT_MIN="1", OMP_NUM_THREADS="4")
    out = subprocess.run([MAKEDB, fasta, str(tmp_path / "par")], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    par = {f: open(str(tmp_path / "par") + f, "rb").read() for f in DB_FILES}
    assert par["0lengths"] == serial["0lengths"] and par["0metadata"] == serial["0metadata"]
    assert len(par["0chars"]) == len(serial["0chars"]) and par["0offsets"] == serial["0offsets"]

    def records(db):
        lengths = np.frombuffer(db["0lengths"], dtype=np.int32)
        offsets = np.frombuffer(db["0offsets"], dtype=np.uint64)
        hoff = np.frombuffer(db["0headeroffsets"], dtype=np.uint64)
        return [(int(lengths[i]), db["0headers"][int(hoff[i]):int(hoff[i + 1])],
                 db["0chars"][int(offsets[i]):int(offsets[i + 1])]) for i in range(len(lengths))]

    rp, rs = records(par), records(serial)
    assert sorted(rp) == sorted(rs)
    # stability: among equal lengths the input order (the header carries the input index) is kept
    headers, _ = O.read_fasta(fasta)
    pos = {h.encode(): i for i, h in enumerate(headers)}
    for (l0, h0, _), (l1, h1, _) in zip(rp, rp[1:]):
        assert l0 < l1 or (l0 == l1 and pos[h0] < pos[h1])


def test_makedb_roundtrip_content(tmp_path):
    """Without the reference: decoded DB content equals the input, sorted by length, padded to 4 with 20."""
    rng = np.random.default_rng(5)
    fasta = str(tmp_path / "in.fa")
    _random_fasta(fasta, rng, 120)
    headers, seqs = O.read_fasta(fasta)
    db = run_makedb(MAKEDB, fasta, str(tmp_path / "db"))
    lengths = np.frombuffer(db["0lengths"], dtype=np.int32)
    offsets = np.frombuffer(db["0offsets"], dtype=np.uint64)
    hoff = np.frombuffer(db["0headeroffsets"], dtype=np.uint64)
    chars = np.frombuffer(db["0chars"], dtype=np.int8)
    assert len(lengths) == len(seqs) and np.all(np.diff(lengths) >= 0) and np.all(offsets % 4 == 0)
    by_header = {h: s for h, s in zip(headers, seqs)}
    for i in range(len(lengths)):
        h = db["0headers"][int(hoff[i]):int(hoff[i + 1])].decode()
        s = by_header[h]
        assert lengths[i] == len(s)
        a = int(offsets[i])
        np.testing.assert_array_equal(chars[a:a + len(s)], O.encode(s))
        assert np.all(chars[a + len(s):int(offsets[i + 1])] == 20)


def test_cli_surface_without_gpu():
    out = subprocess.run([ALIGN, "--help"], capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--query", "--db", "--top", "--gop", "--gex", "--mat", "--maxGpuMem", "--maxTempBytes", "--maxBatchBytes",
                 "--maxBatchSequences", "--dpx", "--of", "--tsv", "--verbose", "--printLengthPartitions", "--interactive",
                 "--prefetchDBFile", "--uploadFull", "--pseudodb", "--singlePassType", "--manyPassType_small",
                 "--manyPassType_large", "--overflowType", "--refCompat"):
        assert flag in out.stdout, flag
    out = subprocess.run([ALIGN, "--db", "x"], capture_output=True, text=True)
    assert "Query is missing" in out.stdout and out.returncode == 0
    out = subprocess.run([ALIGN, "--query", "x"], capture_output=True, text=True)
    assert "DB prefix is missing" in out.stdout
    out = subprocess.run([MAKEDB], capture_output=True, text=True)
    assert "Usage" in out.stdout and "--mem" in out.stdout and "--tempdir" in out.stdout


def test_align_fails_loudly_without_gpu(golden_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = subprocess.run([ALIGN, "--query", os.path.join(golden_dir, "allqueries.fasta"), "--db",
                          os.path.join(golden_dir, "allqueries_db", "aq")], capture_output=True, text=True)
    assert out.returncode != 0 and "No GPU found" in out.stderr


def test_driver_header_symbols_exported():
    header = open(os.path.join(ROOT, "include", "cudasw4_amd_driver.h")).read()
    declared = set(re.findall(r"\b(swdrv_[a-z_0-9]+)\s*\(", header))
    from cudasw4_amd import driver
    assert declared == set(driver.EXPORTS), declared ^ set(driver.EXPORTS)
    lib = ctypes.CDLL(os.path.join(LIBDIR, "libcudasw4_host.so"))
    for name in declared:
        assert hasattr(lib, name), name
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(driver.DriverError):
            driver.Driver()


def test_host_input_helpers_match_reference(golden_dir):
    """The product's own reader / encoder / pseudo-DB generator / matrices (used by bench.py and align)
    against the reference-derived golden data."""
    from cudasw4_amd import driver
    g = O.golden("ref_tables.json")
    headers, seqs = driver.read_sequences(os.path.join(golden_dir, "allqueries.fasta"))
    assert headers == g["query_headers"] and [len(s) for s in seqs] == g["query_lengths"]
    assert driver.encode(bytes(range(256))).tolist() == g["encode_map_256"]
    assert driver.pseudo_sequence(2048, 42).tolist() == g["pseudodb_seed42_first2048"]
    assert driver.pseudo_sequence(64, 7).tolist() == g["pseudodb_seed7_first64"]
    for which in (45, 50, 62, 80):
        assert driver.matrix(which).tolist() == g["blosum21"][str(which)]
    with pytest.raises(ValueError):
        driver.matrix(63)


def restated_shard_ranges(offsets, lengths, world):
    """partitionDBAmongstGpus (cudasw4.cuh:928-1004, dbdata.cpp:265-292) restated independently of the product:
    per length partition <= world contiguous ranges of ~chars/world, each ending at the first subject boundary
    past its quota, leftovers to the last non-empty range."""
    offsets = np.asarray(offsets, dtype=np.uint64).astype(np.int64)
    lengths = np.asarray(lengths, dtype=np.int64)
    ends = np.searchsorted(lengths, O.partition_boundaries(), side="right")
    begins = np.concatenate([[0], ends[:-1]])
    out = [[(int(b), int(b)) for b in begins] for _ in range(world)]
    for p in range(36):
        pb, pe = int(begins[p]), int(ends[p])
        if pe <= pb:
            continue
        quota = int(offsets[pe] - offsets[pb]) // world
        cur = pb
        for r in range(world):
            if cur >= pe:
                break
            if r == world - 1:
                end = pe
            else:
                end = int(np.searchsorted(offsets[cur:pe + 1], int(offsets[cur]) + quota, side="right")) + cur
                end = min(max(end, cur + 1), pe)
            out[r][p] = (cur, end)
            cur = end
        if cur < pe:
            for r in range(world - 1, -1, -1):
                if out[r][p][1] > out[r][p][0]:
                    out[r][p] = (out[r][p][0], pe)
                    break
    return out


def test_corrupt_databases_are_refused(tmp_path):
    """Database::open validates what the kernels rely on: monotonic offsets with room for every padded
    sequence, non-negative ascending lengths, letter codes 0..20, file sizes."""
    dbinspect = os.path.join(LIBDIR, "dbinspect")
    rng = np.random.default_rng(31)
    fasta = str(tmp_path / "in.fa")
    _random_fasta(fasta, rng, 300)
    good = str(tmp_path / "good")
    subprocess.check_call([MAKEDB, fasta, good], stdout=subprocess.DEVNULL)
    assert subprocess.run([dbinspect, good], capture_output=True).returncode == 0

    def variant(name, mutate):
        prefix = str(tmp_path / name)
        for suffix in ("metadata", "0chars", "0offsets", "0lengths", "0headers", "0headeroffsets", "0metadata"):
            open(prefix + suffix, "wb").write(open(good + suffix, "rb").read())
        mutate(prefix)
        p = subprocess.run([dbinspect, prefix], capture_output=True, text=True)
        assert p.returncode == 1, name
        return p.stderr

    def patch(suffix, dtype, fn):
        def m(prefix):
            a = np.fromfile(prefix + suffix, dtype=dtype)
            fn(a)
            a.tofile(prefix + suffix)
        return m

    assert "codes outside" in variant("badcode", patch("0chars", np.int8, lambda a: a.__setitem__(100, 77)))
    assert "codes outside" in variant("negcode", patch("0chars", np.int8, lambda a: a.__setitem__(5, -3)))
    assert "not monotonic" in variant("offdown", patch("0offsets", np.uint64, lambda a: a.__setitem__(10, a[9] - 4)))
    assert "no room" in variant("offtight", patch("0offsets", np.uint64, lambda a: a.__setitem__(slice(200, None), a[200:] - 8)))
    assert "negative" in variant("neglen", patch("0lengths", np.int32, lambda a: a.__setitem__(0, -1)))
    assert "not sorted" in variant("unsorted", patch("0lengths", np.int32, lambda a: a.__setitem__(50, 1)))
    assert "too short" in variant("truncated", lambda prefix: open(prefix + "0chars", "r+b").truncate(1000))
    assert "do not match" in variant("shortoffsets", lambda prefix: open(prefix + "0offsets", "r+b").truncate(8 * 100))


def test_cpp_reader_and_sharding_match_restatement(tmp_path, golden_dir):
    """The C++ DB reader / partition counts / shard_database (the ONE shard cutter: search.shard_ranges calls it too)
    against the reference's metadata file and an independent restatement of the reference's algorithm."""
    import json
    from cudasw4_amd import search
    dbinspect = os.path.join(LIBDIR, "dbinspect")
    rng = np.random.default_rng(29)
    fasta = str(tmp_path / "in.fa")
    _random_fasta(fasta, rng, 1500)
    prefix = str(tmp_path / "db")
    subprocess.check_call([MAKEDB, fasta, prefix], stdout=subprocess.DEVNULL)
    lengths = np.fromfile(prefix + "0lengths", dtype=np.int32)
    offsets = np.fromfile(prefix + "0offsets", dtype=np.uint64)
    meta = open(prefix + "0metadata", "rb").read()
    counts_file = np.frombuffer(meta[4 + 36 * 4:], dtype=np.uint64)
    for shards in (1, 2, 3, 8):
        info = json.loads(subprocess.check_output([dbinspect, prefix, str(shards)]))
        assert info["num_sequences"] == len(lengths) and info["residues"] == int(lengths.sum())
        assert info["partition_counts"] == counts_file.tolist()
        expect = restated_shard_ranges(offsets, lengths, shards)
        assert info["shards"] == [[list(r) for r in shard] for shard in expect]
        assert search.shard_ranges(offsets, lengths, shards) == expect
    info = json.loads(subprocess.check_output([dbinspect, os.path.join(golden_dir, "allqueries_db", "aq"), "2"]))
    assert info["num_sequences"] == 20 and info["first_header"].startswith("gi|")
    bad = subprocess.run([dbinspect, str(tmp_path / "nope")], capture_output=True, text=True)
    assert bad.returncode == 1 and "Cannot open DB" in bad.stderr


def _big_fasta(path, n, seed, mean_len):
    """n sequences with geometric-ish lengths written with numpy (fast): headers s<i>, letters from the 20 + X."""
    rng = np.random.default_rng(seed)
    lengths = np.minimum(1 + rng.geometric(1.0 / mean_len, n), 6 * mean_len).astype(np.int64)
    letters = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVX", dtype=np.uint8)
    with open(path, "wb") as f:
        chunk = 50000
        for b in range(0, n, chunk):
            ls = lengths[b:b + chunk]
            body = letters[rng.integers(0, 21, int(ls.sum()))]
            parts, pos = [], 0
            for i, l in enumerate(ls):
                parts.append(b">s%d\n" % (b + i))
                parts.append(body[pos:pos + l].tobytes())
                parts.append(b"\n")
                pos += int(l)
            f.write(b"".join(parts))
    return lengths


@pytest.mark.parametrize("n,mean_len,parallel", [(300_000, 180, False), (1_100_000, 40, True)])
def test_makedb_at_scale(tmp_path, n, mean_len, parallel):
    """makedb on 3*10^5 sequences (serial sort path: byte-identical to the reference makedb when that is
    built here) and on 1.1*10^6 sequences (above the 2^20 threshold: the parallel stable sort runs for real), under a
    memory limit that makes the five arrays spill to their temp files.  Checked: ascending lengths, offsets = padded
    prefix sums, partition counts, every residue accounted for (per-letter histogram), headers a permutation,
    equal lengths in input order on the parallel path, and the DB loads (dbinspect validates offsets and codes)."""
    fasta = str(tmp_path / "big.fa")
    lengths_in = _big_fasta(fasta, n, 77, mean_len)
    prefix = str(tmp_path / "db")
    env = dict(os.environ, OMP_NUM_THREADS="8")
    out = subprocess.run([MAKEDB, fasta, prefix, "--mem", "48M", "--tempdir", str(tmp_path)], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    lengths = np.fromfile(prefix + "0lengths", dtype=np.int32)
    offsets = np.fromfile(prefix + "0offsets", dtype=np.uint64)
    chars = np.fromfile(prefix + "0chars", dtype=np.int8)
    hoff = np.fromfile(prefix + "0headeroffsets", dtype=np.uint64)
    assert len(lengths) == n and (np.diff(lengths) >= 0).all()
    assert (np.sort(lengths_in) == lengths).all()
    padded = (lengths.astype(np.int64) + 3) // 4 * 4
    assert offsets[0] == 0 and (np.diff(offsets.astype(np.int64)) == padded).all() and len(chars) == int(offsets[-1])
    # per-letter histogram of the whole DB == histogram of the input (+ padding as code 20)
    raw = np.fromfile(fasta, dtype=np.uint8)
    table = O.encode(bytes(range(256)))
    # sequence lines only: drop header lines (they start with '>') — headers are "s<digits>", none of which encodes below 20
    hist_db = np.bincount(chars.astype(np.int64), minlength=21)
    assert hist_db[:20].sum() + hist_db[20] == len(chars)
    seq_letters = int(lengths_in.sum())
    assert hist_db.sum() - (padded.sum() - lengths.astype(np.int64).sum()) == seq_letters
    # headers: a permutation of s0..s(n-1)
    hdr = np.fromfile(prefix + "0headers", dtype=np.uint8).tobytes()
    ids = np.array([int(hdr[int(hoff[i]) + 1:int(hoff[i + 1])]) for i in range(0, n, max(1, n // 5000))])
    assert (lengths_in[ids] == lengths[::max(1, n // 5000)]).all()
    if parallel:
        all_ids = np.array([int(x) for x in hdr.replace(b"s", b" ").split()])
        assert len(all_ids) == n and (np.sort(all_ids) == np.arange(n)).all()
        same = lengths[1:] == lengths[:-1]
        assert (all_ids[1:][same] > all_ids[:-1][same]).all()  # stable: equal lengths keep the input order
    meta = open(prefix + "0metadata", "rb").read()
    counts = np.frombuffer(meta[4 + 36 * 4:], dtype=np.uint64)
    assert counts.sum() == n and counts.tolist() == np.histogram(lengths, bins=np.concatenate([[0], O.partition_boundaries().astype(np.int64) + 1]))[0].tolist()
    info = subprocess.run([os.path.join(LIBDIR, "dbinspect"), prefix, "8"], capture_output=True, text=True)
    assert info.returncode == 0, info.stderr
    if not parallel and os.path.exists(REF_MAKEDB):
        ref = run_makedb(REF_MAKEDB, fasta, str(tmp_path / "ref"))
        for f in DB_FILES:
            assert open(prefix + f, "rb").read() == ref[f], f


def test_vector_loader_fallback_when_the_files_cannot_be_mapped(tmp_path, golden_dir):
    """main.cu:172-191 / dbdata.cpp:118-190: when the DB files cannot be memory-mapped the loader reads them into memory
    (loadDBWithVectors); CUDASW4_AMD_DB_NO_MMAP=1 forces that path.  Same content, same validation."""
    import json
    dbinspect = os.path.join(LIBDIR, "dbinspect")
    prefix = os.path.join(golden_dir, "allqueries_db", "aq")
    a = json.loads(subprocess.check_output([dbinspect, prefix, "3"]))
    b = json.loads(subprocess.check_output([dbinspect, prefix, "3"], env=dict(os.environ, CUDASW4_AMD_DB_NO_MMAP="1")))
    assert a == b and a["num_sequences"] == 20
    bad = subprocess.run([dbinspect, str(tmp_path / "nope")], capture_output=True, text=True,
                         env=dict(os.environ, CUDASW4_AMD_DB_NO_MMAP="1"))
    assert bad.returncode == 1 and "Cannot open DB" in bad.stderr


def test_parallel_blocks_helper(tmp_path):
    """host/parallel_blocks.hpp (the fork-join helper that replaced OpenMP in the host library): every block exactly once
    for any block / thread count, exceptions propagate to the caller."""
    exe = str(tmp_path / "pb_test")
    src = os.path.join(ROOT, "tests", "host", "parallel_blocks_test.cpp")
    inc = os.path.join(ROOT, "cudasw4_amd", "csrc", "host")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I", inc, src, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_residency_plan_hybrid_and_batches():
    """The residency decision of a GPU's shard (plan_residency; GpuWorkingSet / assignBatchesToGpuMem / computeDbCopyPlan,
    cudasw4.cuh:317-392,1087-1144,1177-1277) without a GPU: resident when the chars fit, otherwise the longest subjects
    cached and the rest cut into batches that cover it exactly; the limit is respected; degenerate limits still work."""
    from cudasw4_amd import driver, synthdb
    lengths = synthdb.sprot_like_lengths(40000, seed=5, max_len=12000).astype(np.int64)
    padded = (lengths + 3) // 4 * 4
    off = np.concatenate([[0], np.cumsum(padded)]).astype(np.uint64)
    n, chars, max_len = len(lengths), int(off[-1]), int(lengths[-1])
    meta = 24 * n + 8
    # fits: resident, nothing streamed
    r = driver.plan_residency(off, max_len)
    assert r["cache_begin"] == 0 and r["cache_bytes"] == chars and r["batches"] == []
    # does not fit: hybrid
    for frac in (0.25, 0.5, 0.7):
        batch = 1 << 20
        limit = int((frac * chars + 3 * (batch + 64) + 64) / 0.75) + meta
        r = driver.plan_residency(off, max_len, max_gpu_mem=limit, max_batch_bytes=batch)
        cb = r["cache_begin"]
        assert 0 < cb < n and r["cache_bytes"] == chars - int(off[cb])
        assert abs(r["cache_bytes"] - frac * chars) < 0.02 * chars + max_len + 8      # what the budget allows, to a subject
        assert r["batch_bytes"] == batch
        b = r["batches"]
        assert b[0][0] == 0 and b[-1][1] == cb and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
        assert all(int(off[e] - off[s]) <= batch for s, e in b) and all(e > s for s, e in b)
        # everything the limit pays for: metadata, scratch quarter, staging, cache
        used = meta + (limit - meta) // 4 + 3 * (batch + 64) + r["cache_bytes"]
        assert used <= limit
        # without the cache the same limit streams everything
        r0 = driver.plan_residency(off, max_len, max_gpu_mem=limit, max_batch_bytes=batch, allow_cache=False)
        assert r0["cache_begin"] == n and r0["cache_bytes"] == 0 and r0["batches"][-1][1] == n
    # a limit of one byte: everything streamed, a batch still holds the longest subject; batch sequence limit honoured
    r = driver.plan_residency(off, max_len, max_gpu_mem=1, max_batch_bytes=6000, max_batch_sequences=7)
    assert r["cache_begin"] == n and r["batch_bytes"] >= max_len + 4
    assert all(e - s <= 7 for s, e in r["batches"]) and r["batches"][-1][1] == n
    assert all(int(off[e] - off[s]) <= r["batch_bytes"] for s, e in r["batches"])
    # the 256 MiB safety margin (cudasw4.cuh:1020-1026) applies only to limits above it
    big = driver.plan_residency(off, max_len, max_gpu_mem=(300 << 20) + meta, max_batch_bytes=1 << 20)
    small = driver.plan_residency(off, max_len, max_gpu_mem=(200 << 20) + meta, max_batch_bytes=1 << 20)
    assert small["cache_bytes"] == chars and big["cache_bytes"] == chars   # 23 MB of chars fit both
    # the device's free memory caps the limit
    r = driver.plan_residency(off, max_len, free_mem=chars // 2, max_batch_bytes=1 << 20)
    assert 0 < r["cache_begin"] <= n and r["cache_bytes"] < chars // 2
    # an empty shard
    r = driver.plan_residency(np.zeros(1, dtype=np.uint64), 0, max_gpu_mem=1)
    assert r["cache_begin"] == 0 and r["batches"] == []


def test_residency_plan_budgets_the_scratch_of_every_stream():
    """the limit must cover the stripe-border scratch of EVERY stream that can hold one (work, second work, two
    auxiliary streams, the re-score service — kTempStreams = 5), not one buffer: cached chars + 3 staging buffers + 5 scratch
    buffers at their cap + the per-subject arrays + the safety margin stay inside --maxGpuMem (from 5.25 GiB up, where the quarter
    of the limit behind the safety margin covers the 256 MiB floor of all five)."""
    from cudasw4_amd import driver
    n = 200000
    lengths = np.full(n, 400, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    off = (off * 2000).astype(np.uint64)   # 160 GB of chars: far above every limit below
    meta = 24 * n + 8
    for limit_gb, max_temp in ((5.25, 0), (6, 0), (16, 0), (16, 64 << 20), (64, 0)):
        limit = int(limit_gb * (1 << 30)) + meta
        r = driver.plan_residency(off, 800000, max_gpu_mem=limit, max_batch_bytes=128 << 20, max_temp_bytes=max_temp)
        assert r["cache_begin"] > 0 and r["batches"]
        tps = r["temp_per_stream"]
        assert tps >= min(max_temp or (4 << 30), 256 << 20) and tps <= (max_temp or (4 << 30))
        used = meta + (256 << 20) + r["cache_bytes"] + 64 + 3 * (r["batch_bytes"] + 64) + 5 * tps
        assert used <= limit, (limit_gb, max_temp, used - limit)
    # resident with memory to spare: every buffer may grow to --maxTempBytes
    small = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    assert driver.plan_residency(small, 400)["temp_per_stream"] == 4 << 30
    assert driver.plan_residency(small, 400, max_temp_bytes=1 << 20)["temp_per_stream"] == 1 << 20
    # resident, but barely: the buffers share what is left (never below the floor)
    tight = driver.plan_residency(small, 400, max_gpu_mem=int(small[-1]) + (2 << 30) + meta)
    assert tight["cache_begin"] == 0 and (256 << 20) <= tight["temp_per_stream"] <= (2 << 30) // 4 + (1 << 20)
