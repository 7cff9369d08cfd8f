"""GPU: BASELINE.json configs 3 and 5 on their own workloads, and bench.py's multi-rank path.

config 3  allqueries vs a Swiss-Prot-sized DB, packed-int16 kernels (runsprotbenchmark.sh:42-44 `--dpx`)
config 5  a DB above the memory limit: batch streaming per GPU with the int32 kernels everywhere
          (runtremblbenchmark.sh, cudasw4.cuh:1560-1712), several shards in flight at once
bench     `bench.py --gpus 2` (two ranks on this box's one GPU through the gloo test hooks): strong scaling shards
          ONE DB, the merged top-K equals the 1-rank run
Everything goes through the C++ host driver -> C ABI -> HIP kernels and is compared with the CPU oracle.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = O.ROOT
FASTA = os.path.join(O.GOLDEN_DIR, "allqueries.fasta")


def sample_db(chars, offsets, lengths, pick):
    from cudasw4_amd import search
    return search.build_shard(chars, offsets, lengths, [(int(i), int(i) + 1) for i in pick])[:3]


@pytest.fixture(scope="module")
def sprot_db():
    from cudasw4_amd import synthdb
    return synthdb.sprot_like()  # 570 000 sequences, ~2e8 residues, seeded


def test_config3_sprot_like_dpx_all_queries(sprot_db):
    """Config 3 at full size: every one of the 20 queries against the 570 k-sequence Swiss-Prot-like DB with the
    --dpx kernel configuration.  (i) every score of a seeded 2 000-subject sample that includes the whole > 8000
    tail equals the oracle; (ii) all 570 000 scores equal those of the all-int32 configuration (an independent
    arithmetic path); (iii) the top-10 equals the top-10 of all scores, whose entries are oracle-checked."""
    from cudasw4_amd import driver, search
    chars, offsets, lengths = sprot_db
    n = len(lengths)
    _, letters = O.read_fasta(FASTA)
    queries = [O.encode(q) for q in letters]
    rng = np.random.default_rng(5)
    tail = np.nonzero(lengths > 8000)[0]
    assert len(tail) >= 10 and lengths[-1] > 35000
    pick = np.unique(np.concatenate([rng.choice(n, 2000, replace=False), tail]))
    sub = sample_db(chars, offsets, lengths, pick)

    d16 = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2))
    d16.db_from_arrays(chars, offsets, lengths)
    d16.upload()
    assert d16.shard_info(0)["resident"] and d16.shard_info(0)["subjects"] == n
    d32 = driver.Driver(devices=[0], num_top=0, kinds=(2, 1, 2, 2))
    d32.db_from_arrays(chars, offsets, lengths)
    d32.upload()
    total_ovf = total_rescored = 0
    for qi, q in enumerate(letters):
        r = d16.scan(q)
        sc, ids = d16.last_scores(0)
        assert (ids == np.arange(n)).all()
        expect = O.scan(queries[qi], *sub, simd=True)
        assert (sc[pick] == expect).all(), (qi, np.nonzero(sc[pick] != expect)[0][:5])
        d32.scan(q)
        sc32, _ = d32.last_scores(0)
        assert (sc == sc32).all(), (qi, np.nonzero(sc != sc32)[0][:5])
        es, ei = search.merge_topk([(sc, ids)], 10)
        assert r["scores"].tolist() == es.tolist() and r["ids"].tolist() == ei.tolist(), qi
        # the winners themselves against the oracle
        top_sub = sample_db(chars, offsets, lengths, np.sort(ei))
        assert sorted(O.scan(queries[qi], *top_sub, simd=True).tolist(), reverse=True) == es.tolist()
        total_ovf += r["num_overflows"]
        total_rescored += r["num_rescored"]
        assert r["num_rescored"] >= r["num_overflows"] == int((sc >= 25000).sum())
        assert r["gcups"] > 0
    # the DB holds the queries themselves and their relatives (synthdb.family_members): the packed-int16 launches flag
    # subjects and the 32-bit kind re-scores them — the load a real Swiss-Prot puts on config 3
    assert total_rescored > 0 and total_ovf >= 3
    d16.close()
    d32.close()


@pytest.mark.parametrize("kinds", [(2, 1, 2, 2), (1, 1, 2, 2)])
def test_config5_streaming_int32_three_shards(kinds):
    """Config 5's route: forced batch streaming (memory limit below the shard size), three shards in flight
    (devices=[0,0,0]: one worker thread per shard), int32 kernels for every single-pass partition, batches that span
    length partitions — vs the oracle on a ragged DB with all partition classes incl. > 8000."""
    from cudasw4_amd import driver, synthdb
    lengths = synthdb.sprot_like_lengths(24000, seed=11, max_len=12000)
    chars, offsets, lengths = synthdb.random_db(lengths, seed=12, other_fraction=0.01)
    _, letters = O.read_fasta(FASTA)
    d = driver.Driver(devices=[0, 0, 0], num_top=25, kinds=kinds, max_gpu_mem=1, max_batch_bytes=400_000)
    d.db_from_arrays(chars, offsets, lengths)
    assert d.num_gpus() == 3 and not any(d.shard_info(g)["resident"] for g in range(3))
    assert sum(d.shard_info(g)["subjects"] for g in range(3)) == len(lengths)
    for qi in (0, 7, 13, 19):
        r = d.scan(letters[qi])
        expect = O.scan(O.encode(letters[qi]), chars, offsets, lengths, simd=True)
        ids, sc = d.all_scores()
        got = np.empty_like(sc)
        got[ids] = sc
        assert (got == expect).all(), (kinds, qi, np.nonzero(got != expect)[0][:5])
        es, ei = O.topk(expect, 25)
        assert r["scores"].tolist() == es.tolist() and r["ids"].tolist() == ei.tolist()
        iv = d.batch_intervals()
        assert len(iv) >= 3 * 5  # every shard needed several batches
        if kinds[0] == 2:
            assert r["num_overflows"] <= int((expect >= 25000 - 12500).sum())  # only partition 34 is packed
    # (that the three shards' spans overlap in time is asserted in tests/test_gpu_zz_timing.py, behind every parity test)
    assert len(d.gpu_spans()) == 3
    d.close()


def test_streaming_pinned_fallback_matches(monkeypatch):
    """The pinned-staging fallback of the streamed path (no hipHostRegister of the DB mapping) gives the same result."""
    from cudasw4_amd import driver, synthdb
    lengths = synthdb.sprot_like_lengths(6000, seed=3, max_len=9000)
    chars, offsets, lengths = synthdb.random_db(lengths, seed=4)
    _, letters = O.read_fasta(FASTA)
    expect = O.scan(O.encode(letters[5]), chars, offsets, lengths, simd=True)
    for env in ("0", "1"):
        monkeypatch.setenv("CUDASW4_AMD_NO_HOSTREGISTER", env)
        d = driver.Driver(devices=[0], num_top=5, kinds=(0, 0, 3, 3), max_gpu_mem=1, max_batch_bytes=200_000)
        d.db_from_arrays(chars, offsets, lengths)
        r = d.scan(letters[5])
        ids, sc = d.all_scores()
        assert (sc[np.argsort(ids)] == expect).all(), env
        es, ei = O.topk(expect, 5)
        assert r["scores"].tolist() == es.tolist() and r["ids"].tolist() == ei.tolist()
        d.close()


def run_bench(extra, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, env=env,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_on_one_gpu_strong_and_weak():
    """`bench.py --gpus 2` spawns its two ranks itself; with the test hooks both use GPU 0 and gloo.  Strong scaling
    (default): ONE DB is sharded, the merged top-10 equals the 1-rank run and every score is verified; weak: one DB
    per rank, value counts both."""
    common = ["--steps", "1", "--warmup", "0", "--db-size", "200000", "--no-cpu-baseline", "--no-secondary"]
    one = run_bench(["--gpus", "1"] + common)
    assert one["n_gpus"] == 1 and one["verified"] is True and one["scaling"] == "single" and "sprot_like" not in one
    assert one["roofline"]["launches"] >= 1 and one["roofline"]["avg_launch_ms"] > 0
    hooks = {"BENCH_FORCE_DEVICE": "0", "BENCH_DIST_BACKEND": "gloo"}
    two = run_bench(["--gpus", "2"] + common, hooks)
    assert two["n_gpus"] == 2 and two["verified"] is True and two["scaling"] == "strong"
    assert two["config"]["db_subjects"] == 200000
    assert two["config"]["top_merged_example"] == one["config"]["top_merged_example"] and "weak_scaling" not in two
    weak = run_bench(["--gpus", "2", "--scaling", "weak"] + common, hooks)
    assert weak["n_gpus"] == 2 and weak["verified"] is True and weak["scaling"] == "weak"
    assert weak["config"]["db_subjects"] == 400000


def test_bench_eight_ranks_on_one_gpu_default_line():
    """What the first real 8-GPU scaling run will execute — `bench.py --gpus 8`, default flags: eight ranks (here all on
    GPU 0, gloo), ONE 10^6 x 512 DB in eight shards, every score of every rank verified, the merged top-10 equal to the
    1-rank line's, the weak-scaling run and the Swiss-Prot-like workload in the same line."""
    hooks = {"BENCH_FORCE_DEVICE": "0", "BENCH_DIST_BACKEND": "gloo"}
    one = run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"])
    out = run_bench(["--gpus", "8", "--steps", "1", "--warmup", "0"], hooks)
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["verified"] is True
    assert out["config"]["db_subjects"] == 1_000_000
    assert out["config"]["top_merged_example"] == one["config"]["top_merged_example"]
    assert out["weak_scaling"]["verified"] is True and out["weak_scaling"]["config"]["db_subjects"] == 8_000_000
    assert out["sprot_like"]["verified"] is True and out["sprot_like"]["config"]["db_subjects"] == 570000
    assert out["cpu_baseline"] is None  # rank 0 times the CPU leg at N = 1 only


def test_bench_rank_failure_fails_the_job():
    """A rank that dies takes the job down: the parent (which started the ranks as fresh child processes before touching
    the GPU) exits non-zero and prints no result line."""
    env = dict(os.environ, BENCH_FORCE_DEVICE="0", BENCH_DIST_BACKEND="gloo", BENCH_FAIL_RANK="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0",
                        "--db-size", "50000", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_two_rank_default_line_carries_weak_scaling_and_sprot_like():
    """The default multi-rank line: strong scaling headline + the weak-scaling run + the Swiss-Prot-like workload."""
    hooks = {"BENCH_FORCE_DEVICE": "0", "BENCH_DIST_BACKEND": "gloo"}
    out = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--db-size", "100000", "--no-cpu-baseline"], hooks)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["verified"] is True
    assert out["weak_scaling"]["scaling"] == "weak" and out["weak_scaling"]["verified"] is True
    assert out["weak_scaling"]["config"]["db_subjects"] == 200000 and out["sprot_like"]["verified"] is True


def test_bench_process_group_path_with_the_real_backend():
    """The torch.distributed code path of bench.py with the backend the multi-GPU runs use (nccl == RCCL): process-group
    init on the device, the per-step gather of CUDA tensors, the reductions and barriers — with the one rank a 1-GPU box
    can give it (two RCCL ranks cannot share a GPU), under torch.distributed.run exactly as the round driver launches it."""
    env = dict(os.environ, BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
           "--db-size", "100000", "--no-cpu-baseline", "--no-secondary"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["verified"] is True and out["value"] > 0
    assert out["config"]["top_merged_example"]["ids"] == list(range(10))


def test_bench_default_line_carries_the_sprot_like_secondary():
    """The default command (what the round driver runs) reports config 3 next to the headline: same JSON line, own
    timed region, every score verified (here without the CPU leg: packed vs int32 configuration)."""
    out = run_bench(["--steps", "1", "--warmup", "0", "--db-size", "100000", "--no-cpu-baseline"])
    assert out["verified"] is True and out["dtype"] == "f16x2"
    sec = out["sprot_like"]
    assert sec["verified"] is True and sec["dtype"] == "i16x2" and sec["value"] > 0 and sec["config"]["db_subjects"] == 570000
    assert sec["roofline"]["launches"] >= 1
    # the Swiss-Prot-like leg carries overflow / re-score load, and says how much
    ol = sec["overflow_load"]
    assert ol["num_rescored"] >= ol["num_overflows"] >= 3 and ol["rescore_launches_per_step"] >= 20 and ol["rescore_ms_per_step"] > 0
    assert out["overflow_load"]["num_rescored"] == 0        # the peak DB's scores are all below 60
    # the whole peak protocol (runpeakbenchmark.sh:26-83) is in the line, every cell verified
    sw = out["peak_sweep"]
    assert sw["verified"] is True
    assert sorted(sw["gcups"]) == ["dpxs16", "dpxs32", "float", "half2"]
    assert sorted(map(int, sw["gcups"]["half2"])) == [128, 256, 512, 768, 1024, 2048] == sorted(map(int, sw["gcups"]["dpxs16"]))
    assert sorted(map(int, sw["gcups"]["float"])) == [128, 256, 512, 768, 1024] == sorted(map(int, sw["gcups"]["dpxs32"]))
    assert all(v > 0 for row in sw["gcups"].values() for v in row.values())


def test_bench_sprot_like_workload_small():
    """`bench.py --workload sprot-like` (config 3's line for the driver's clock) on a reduced DB: CPU leg present,
    every sampled score verified against the oracle, roofline from live HIP events."""
    out = run_bench(["--workload", "sprot-like", "--db-size", "60000", "--steps", "1", "--warmup", "0",
                     "--cpu-sample-subjects", "400"])
    assert out["verified"] is True and out["dtype"] == "i16x2" and out["cpu_baseline"]["value"] > 0
    assert out["roofline"]["achieved"] > 0 and out["value"] > 0


def test_bench_trembl_like_reduced_leg():
    """BASELINE config 5's shape at a tenth of a tenth of TrEMBL (10^7 sequences, 3.8e9 residues generated on the device;
    the full 2.5e8-sequence legs are in profiles/r05_scale/): the int32 configuration under a memory limit far below the DB
    — hybrid residency, most of the chars streamed on every query — two queries, every sampled score equal to the CPU
    oracle, top-10 equal to the top of all 10^7 scores, the H2D counter at the streamed part's size."""
    out = run_bench(["--workload", "trembl-like", "--db-size", "10000000", "--kernel", "dpxs32", "--queries", "0,9", "--steps", "1",
                     "--warmup", "0", "--max-gpu-mem", "2G", "--no-secondary", "--no-sweep", "--cpu-sample-subjects", "600"])
    c = out["config"]
    assert out["verified"] is True and c["db_subjects"] == 10_000_000 and c["residency"] == "hybrid"
    assert 0 < c["cached_chars"] < c["shard_chars"]
    assert abs(c["h2d_subject_bytes_per_step"] - 2 * (c["shard_chars"] - c["cached_chars"])) <= 0.03 * c["shard_chars"]   # (+ the next scan's first batch, staged ahead)
    assert out["cpu_baseline"]["value"] > 0 and out["value"] > 1000


def test_bench_real_db_switch(tmp_path):
    """`bench.py --db-prefix P` / $CUDASW4_SPROT_PREFIX (runsprotbenchmark.sh:18-51): the sprot-like leg runs on a real
    DB made by `makedb` — here a 50 000-sequence FASTA — instead of the synthetic stand-in, says so in `data`, and
    its scores are verified against the CPU oracle on a sample of the same files."""
    from cudasw4_amd import driver, synthdb
    lengths = synthdb.sprot_like_lengths(50000, seed=41, max_len=12000)
    chars, offsets, lengths = synthdb.random_db(lengths, seed=42, other_fraction=0.005)
    alphabet = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVX", dtype=np.uint8)
    fasta = tmp_path / "db.fasta"
    rng = np.random.default_rng(43)
    order = rng.permutation(len(lengths))   # makedb sorts by length itself
    with open(fasta, "wb") as f:
        for i in order:
            s = alphabet[chars[int(offsets[i]):int(offsets[i]) + int(lengths[i])]]
            f.write(b">seq%d some header\n" % i)
            f.write(s.tobytes() + b"\n")
    prefix = str(tmp_path / "realdb")
    p = subprocess.run([driver.MAKEDB, str(fasta), prefix], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = run_bench(["--workload", "sprot-like", "--db-prefix", prefix, "--steps", "1", "--warmup", "0", "--cpu-sample-subjects", "300"])
    assert out["data"] == "real" and out["verified"] is True and out["config"]["db_subjects"] == 50000
    assert out["config"]["db_residues"] == int(lengths.astype(np.int64).sum()) and prefix in out["config"]["workload"]
    assert out["cpu_baseline"]["value"] > 0 and out["dtype"] == "i16x2"
    # the default command picks the prefix up from the environment for its secondary leg
    out = run_bench(["--steps", "1", "--warmup", "0", "--db-size", "100000", "--no-cpu-baseline"], {"CUDASW4_SPROT_PREFIX": prefix})
    assert out["data"] == "synthetic" and out["sprot_like"]["data"] == "real" and out["sprot_like"]["verified"] is True
    assert out["sprot_like"]["config"]["db_subjects"] == 50000
