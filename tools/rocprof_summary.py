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
    """profiles/kernel_counters.json for bench.py, from one profiled pass of a bench.py command
    (`--steps 1 --warmup 0 --no-verify --no-cpu-baseline --no-secondary --kernel-table`: exactly one pass, and the line
    carries the per-kernel table of that pass):

    * valu_instr_per_unit[workload:kernel configuration:residency]: VALU lane-instructions per useful cell (pair) over
      the pass = SQ_INSTS_VALU x 64 over all DP launches / cells of the pass;
    * traffic_bytes_per_char[workload|kernel instantiation]: HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE, in KB, corrected as
      MI355X_MICROARCH.md prescribes for gfx950) over all launches of the instantiation / the subject bytes those
      launches read (from the line's table) — bench.py scales it by the subject bytes of ITS launches.

    Every entry names its own source file; the whole file is tied to the sha of the kernel sources."""
    import hashlib
    import json
    import os
    import re
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    line = [l for l in open(bench_log).read().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    workload = b["config"]["workload"].split(":")[0]
    kernel_cfg = b["config"]["kernel"]
    residency = b["config"]["residency"]
    native = ":i32native" if os.environ.get("CUDASW4_AMD_I32_NATIVE") == "1" else ""
    packed = b["dtype"].split()[0] in ("f16x2", "i16x2")
    cells = sum(k["cells"] for k in b["kernels"])
    db = sqlite3.connect(valu_db)
    insts = sum(v for (v,) in db.execute(
        "select value from counters_collection where counter_name = 'SQ_INSTS_VALU' and (kernel_name like '%sw_scan_kernel%' or kernel_name like '%sw_scan_stream_kernel%')"))
    ipu = insts * 64.0 / (cells / (2 if packed else 1))
    # KERNEL_COUNTERS_OUT: where the file is built (tools/collect_all_profiles.sh builds it under gpurun_out/ and only
    # replaces the tracked profiles/kernel_counters.json once every expected entry is there)
    out_path = os.environ.get("KERNEL_COUNTERS_OUT") or os.path.join(root, "profiles", "kernel_counters.json")
    try:
        cur = json.load(open(out_path))
    except (OSError, ValueError):
        cur = {}
    sys.path.insert(0, root)
    import bench  # the ONE list of kernel sources and the one hash over them (bench.py: KERNEL_SOURCES, kernel_source_sha)
    sha = bench.kernel_source_sha()
    if cur.get("kernel_src_sha16") != sha or "traffic_bytes_per_char" not in cur:
        cur = {"kernel_src_sha16": sha, "valu_instr_per_unit": {}, "traffic_bytes_per_char": {}}
    key = "%s:%s:%s%s" % (workload, kernel_cfg, residency, native)
    cur["valu_instr_per_unit"][key] = {"value": round(ipu, 3), "unit": "lane-instructions per cell pair" if packed else "lane-instructions per cell",
                                       "source": source, "commit": commit}

    def sum_per_kernel(path, counter):
        acc = defaultdict(lambda: [0, 0.0])
        for name, value in sqlite3.connect(path).execute(
                "select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            # (the streamed kernels of the packed kinds, sw_scan_stream_kernel<kind, R, lanes, multi>, count with the
            # sw_scan_kernel instantiation of the same shape: bench.py's table has one row per (kind, R, lanes, multi))
            name = re.sub(r"sw_scan_stream_kernel<(\d+), (\d+), (\d+), (true|false)>", r"sw_scan_kernel<\1, \2, \3, \4, true>", name)
            if "sw_scan_kernel" in name:
                acc[name][0] += 1
                acc[name][1] += value
        return acc

    f = sum_per_kernel(fetch_db, "FETCH_SIZE")
    w = sum_per_kernel(write_db, "WRITE_SIZE")
    table = {k["kernel"].replace(" *>", ""): k for k in b["kernels"]}
    detail = {}
    for name in f:
        m = re.search(r"(sw_scan_kernel<\d+, \d+, \d+, (?:true|false),)", name)
        if not m or m.group(1) not in table:
            continue
        t = table[m.group(1)]
        if t["launches"] != f[name][0] or t["chars"] <= 0:
            continue  # another form of the recurrence shares the prefix, or the passes differ: no figure rather than a wrong one
        # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE under-reports by 2x on gfx950
        total = (2.0 * f[name][1] + (w[name][1] if name in w else 0.0)) * 1024
        tkey = "%s|%s" % (workload, m.group(1))
        old = cur["traffic_bytes_per_char"].get(tkey)
        if old and old.get("chars_per_launch", 0) > t["chars"] / t["launches"]:
            continue  # the figure of the larger launches (the resident run) stays: batches only add launch-edge traffic
        cur["traffic_bytes_per_char"][tkey] = {"value": round(total / t["chars"], 4), "nstripes": t["nstripes"],
                                                     "launches": t["launches"], "chars_per_launch": int(t["chars"] / t["launches"]),
                                                     "source": source, "commit": commit}
        detail[name] = {"launches": f[name][0], "fetch_kb_raw_sum": f[name][1], "write_kb_sum": w[name][1] if name in w else 0.0,
                        "traffic_bytes": int(total), "subject_bytes": t["chars"]}
    json.dump(cur, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({"key": key, "valu_instr_per_unit": round(ipu, 3), "valu_wave_instructions": insts, "cells": cells,
                      "per_kernel_traffic": detail}, indent=1))


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
