"""CPU: bench.py's launcher logic, the one launch planner / shard cutter shared by the C++ driver and the Python
mirror, and the synthetic DB generator.  No GPU, no compute calls."""
import importlib.util
import os
import subprocess
import sys

import numpy as np

import oracle_lib as O

ROOT = O.ROOT


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_spawn_command_and_defaults():
    b = load_bench()
    a = b.parse_args(["--gpus", "8", "--steps", "3", "--warmup", "1"])
    assert a.gpus == 8 and a.scaling == "strong" and a.workload == "peak" and a.top == 10
    cmd = b.spawn_command(8, ["--gpus", "8", "--steps", "3"], 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert b.kinds_for(b.parse_args([]))[1] == (0, 0, 3, 3)
    assert b.kinds_for(b.parse_args(["--workload", "sprot-like"]))[1] == (1, 1, 2, 2)
    assert b.kinds_for(b.parse_args(["--kernel", "dpxs32"]))[1] == (2, 1, 2, 2)
    assert b.parse_size("1G") == 1 << 30 and b.parse_size("0") == 0


def test_bench_refuses_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, env=env)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr


def test_bench_counters_are_tied_to_the_kernel_sources():
    b = load_bench()
    c, note = b.load_counters()
    assert (c is None) != (note is None)
    if c is not None:
        assert c["kernel_src_sha16"] == b.kernel_source_sha()


def reference_walk(lengths, kinds):
    """The partition walk restated independently (cudasw4.cuh:1742-2103 + merging of equal kinds)."""
    bounds = O.partition_boundaries()
    ends = np.searchsorted(lengths, bounds, side="right")
    begins = np.concatenate([[0], ends[:-1]])
    runs = []
    for p in range(35, -1, -1):
        b, e = int(begins[p]), int(ends[p])
        if e <= b:
            continue
        kind = kinds[0] if p < 34 else kinds[1] if p == 34 else kinds[2]
        if runs and runs[-1]["kind"] == kind and runs[-1]["begin"] == e and (runs[-1]["part_id"] >= 34) == (p >= 34):
            runs[-1]["begin"] = b
        else:
            runs.append({"kind": kind, "part_id": p, "begin": b, "end": e, "maxlen": int(lengths[e - 1])})
    return runs


def test_one_planner_for_both_host_drivers():
    from cudasw4_amd import driver, search
    rng = np.random.default_rng(0)
    for trial in range(20):
        n = int(rng.integers(1, 400))
        lengths = np.sort(rng.choice([3, 48, 49, 64, 65, 200, 256, 257, 512, 700, 1280, 1281, 5000, 8000, 8001, 20000], n)).astype(np.int32)
        for kinds in ((0, 0, 3), (1, 1, 2), (2, 1, 2), (3, 0, 3)):
            got = driver.plan_runs(lengths, *kinds)
            assert got == reference_walk(lengths, kinds), (trial, kinds)
    assert driver.plan_runs(np.zeros(0, np.int32), 0, 0, 3) == []
    # the Python mirror's merged plan IS the C++ planner's
    import inspect
    assert "driver.plan_runs" in inspect.getsource(search.Searcher._launch_plan)


def test_shard_ranges_cover_and_balance():
    from cudasw4_amd import driver, synthdb
    lengths = synthdb.sprot_like_lengths(20000, seed=1, max_len=9000)
    chars, offsets, lengths = synthdb.random_db(lengths, seed=2)
    for world in (1, 2, 3, 8):
        r = driver.shard_ranges(offsets, lengths, world)
        assert len(r) == world and all(len(x) == 36 for x in r)
        covered = np.zeros(len(lengths), dtype=np.int32)
        per_rank_chars = []
        for rank in range(world):
            c = 0
            for (b, e) in r[rank]:
                covered[b:e] += 1
                c += int(offsets[e] - offsets[b])
            per_rank_chars.append(c)
        assert (covered == 1).all()
        if world > 1:
            assert max(per_rank_chars) < 1.25 * (sum(per_rank_chars) / world)


def test_synthetic_sprot_like_db_layout():
    from cudasw4_amd import synthdb
    l = synthdb.sprot_like_lengths(30000)
    assert (np.diff(l) >= 0).all() and l[0] >= 2 and l[-1] == synthdb.SPROT_MAX_LENGTH
    chars, offsets, lengths = synthdb.random_db(l[:3000], seed=9, other_fraction=0.05)
    assert offsets[0] == 0 and len(offsets) == 3001 and (np.diff(offsets.astype(np.int64)) == (lengths.astype(np.int64) + 3) // 4 * 4).all()
    for i in (0, 17, 2999):
        s = chars[int(offsets[i]):int(offsets[i + 1])]
        assert (s[lengths[i]:] == 20).all() and s[:lengths[i]].max() <= 20 and s.min() >= 0
    a = synthdb.sprot_like(2000)
    b = synthdb.sprot_like(2000)
    assert all((x == y).all() for x, y in zip(a, b))
