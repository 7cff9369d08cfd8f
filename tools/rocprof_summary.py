#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into small text files for profiles/.

    python tools/rocprof_summary.py stats  <results.db>            -> per-kernel calls / total / avg (ns)
    python tools/rocprof_summary.py pmc    <results.db> [filter]   -> per-kernel counter sums and per-dispatch means
"""
import sqlite3
import sys
from collections import defaultdict


def stats(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("%-78s %6s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows:
        print("%-78s %6d %14.0f %14.0f %7.2f" % (name[:78], calls, total, avg, pct))
    print()
    print("per-dispatch register/LDS use of the DP kernels:")
    q = ("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x, count(*), avg(duration) "
         "from kernels where name like '%sw_scan_kernel%' group by name, grid_x order by avg(duration) desc")
    print("%-60s %5s %5s %5s %7s %8s %5s %6s %12s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "grid", "wg", "calls", "avg_ns"))
    for r in db.execute(q):
        print("%-60s %5d %5d %5d %7d %8d %5d %6d %12.0f" % ((r[0][:60],) + tuple(r[1:])))


def pmc(path, flt="swk::sw_s"):
    db = sqlite3.connect(path)
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for name, counter, value, dur in db.execute(
            "select kernel_name, counter_name, value, duration from counters_collection"):
        if flt and flt not in name:
            continue
        a = acc[(name, counter)]
        a[0] += 1
        a[1] += value
        a[2] += dur
    print("%-60s %-24s %6s %18s %18s %14s" % ("kernel", "counter", "calls", "sum", "mean/dispatch", "avg_ns"))
    for (name, counter), (n, s, d) in sorted(acc.items()):
        print("%-60s %-24s %6d %18.1f %18.1f %14.0f" % (name[:60], counter, n, s, s / n, d / n))


def traffic(fetch_db, write_db):
    """HBM traffic per launch of the dominant DP kernel, corrected as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE (KB) under-reports wide coalesced reads by 2x on gfx950 -> doubled; WRITE_SIZE (KB) as is."""
    import json

    def mean_per_kernel(path, counter):
        db = sqlite3.connect(path)
        acc = defaultdict(lambda: [0, 0.0, 0.0])
        for name, value, dur in db.execute(
                "select kernel_name, value, duration from counters_collection where counter_name = ?", (counter,)):
            if "sw_scan_kernel" in name:
                a = acc[name]
                a[0] += 1
                a[1] += value
                a[2] += dur
        return acc

    f = mean_per_kernel(fetch_db, "FETCH_SIZE")
    w = mean_per_kernel(write_db, "WRITE_SIZE")
    dominant = max(f, key=lambda k: f[k][2])
    fetch_kb = f[dominant][1] / f[dominant][0]
    write_kb = w[dominant][1] / w[dominant][0] if dominant in w else 0.0
    print(json.dumps({"kernel": dominant, "launches": f[dominant][0], "fetch_kb_raw_mean": fetch_kb,
                      "write_kb_mean": write_kb, "fetch_correction": 2.0,
                      "traffic_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024),
                      "avg_launch_ns_under_pmc": f[dominant][2] / f[dominant][0]}))


def counters(bench_log, fetch_db, write_db, valu_db, source, commit):
    """profiles/kernel_counters.json for bench.py: VALU lane-instructions per useful cell (pair) over one pass of the
    workload (SQ_INSTS_VALU x 64 over all DP launches / cells of the pass) and the HBM traffic per launch of the dominant
    DP kernel, tied to the sha of the kernel sources they were measured on.  `bench_log` holds the JSON line of the
    profiled command (bench.py --steps 1 --warmup 0 --no-verify --no-cpu-baseline: exactly one pass)."""
    import hashlib
    import json
    import os
    import re
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    line = [l for l in open(bench_log).read().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    workload = b["config"]["workload"].split(":")[0]
    dtype = b["dtype"]
    sum_q = int(re.search(r"\((\d+) queries, (\d+) residues\)", b["config"]["workload"]).group(2))
    cells = float(sum_q) * float(b["config"]["db_residues"]) * (b["steps"] + b["warmup"])
    packed = dtype in ("f16x2", "i16x2")
    db = sqlite3.connect(valu_db)
    insts = sum(v for (v,) in db.execute(
        "select value from counters_collection where counter_name = 'SQ_INSTS_VALU' and kernel_name like '%sw_scan_kernel%'"))
    ipu = insts * 64.0 / (cells / (2 if packed else 1))
    out_path = os.path.join(root, "profiles", "kernel_counters.json")
    try:
        cur = json.load(open(out_path))
    except (OSError, ValueError):
        cur = {}
    h = hashlib.sha256()
    for rel in ("cudasw4_amd/csrc/sw_dp_kernel.hpp", "cudasw4_amd/csrc/sw_launch.hpp", "cudasw4_amd/csrc/sw_api.hip", "cudasw4_amd/csrc/Makefile"):
        h.update(open(os.path.join(root, rel), "rb").read())
    sha = h.hexdigest()[:16]
    if cur.get("kernel_src_sha16") != sha:
        cur = {"kernel_src_sha16": sha, "valu_instr_per_unit": {}, "traffic_bytes_per_launch": {}}
    cur["commit"] = commit
    cur["source"] = source
    cur["valu_instr_per_unit"]["%s:%s" % (workload, dtype)] = round(ipu, 3)

    def mean_per_kernel(path, counter):
        acc = defaultdict(lambda: [0, 0.0, 0.0])
        for name, value, dur in sqlite3.connect(path).execute(
                "select kernel_name, value, duration from counters_collection where counter_name = ?", (counter,)):
            if "sw_scan_kernel" in name:
                a = acc[name]
                a[0] += 1
                a[1] += value
                a[2] += dur
        return acc

    f = mean_per_kernel(fetch_db, "FETCH_SIZE")
    w = mean_per_kernel(write_db, "WRITE_SIZE")
    detail = {}
    for name in f:
        m = re.search(r"sw_scan_kernel<(\d+), (\d+), (\d+), (true|false), (true|false)>", name)
        if not m:
            continue
        kind = ["f16x2", "i16x2", "i32", "f32"][int(m.group(1))]
        fetch_kb = f[name][1] / f[name][0]
        write_kb = w[name][1] / w[name][0] if name in w else 0.0
        # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE under-reports by 2x on gfx950
        key = "%s:%s:R%s" % (workload, kind, m.group(2))
        tr = int((2.0 * fetch_kb + write_kb) * 1024)
        if m.group(3) == "16":
            cur["traffic_bytes_per_launch"][key] = tr
        detail[name] = {"launches": f[name][0], "fetch_kb_raw_mean": fetch_kb, "write_kb_mean": write_kb, "traffic_bytes_per_launch": tr,
                        "avg_launch_ns_under_pmc": f[name][2] / f[name][0]}
    json.dump(cur, open(out_path, "w"), indent=1)
    print(json.dumps({"kernel_counters": cur, "valu_wave_instructions": insts, "cells": cells, "per_kernel_traffic": detail}, indent=1))


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    if mode == "counters":
        counters(*sys.argv[2:8])
        sys.exit(0)
    if mode == "stats":
        stats(path)
    elif mode == "traffic":
        traffic(sys.argv[2], sys.argv[3])
    else:
        pmc(path, sys.argv[3] if len(sys.argv) > 3 else "swk::sw_s")
