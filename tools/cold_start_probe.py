#!/usr/bin/env python3
"""Cold first pass of the 20 queries through the C++ driver (no warm-up scans), optionally after keeping the GPU busy
for a while: where do the 70-80 ms stalls of a cold `align` run come from?  Each variant runs in a fresh process.

  python tools/cold_start_probe.py            # parent: spawns the variants, prints per-query ms
  python tools/cold_start_probe.py child MODE # one variant"""
import os
import subprocess
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def child(mode):
    sys.path.insert(0, ROOT)
    import torch
    from cudasw4_amd import driver
    _, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
    d = driver.Driver(devices=[0], num_top=0, kinds=(0, 0, 3, 3))
    d.pseudo_db(1_000_000, 512)
    d.upload()
    if mode.startswith("busy"):
        ms = int(mode[4:])
        a = torch.randn(4096, 4096, device="cuda")
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            (a @ a).sum().item()
    elif mode.startswith("sleep"):
        time.sleep(int(mode[5:]) / 1e3)
    elif mode == "scan0twice":
        d.scan(letters[0])
    out = []
    t_all = time.perf_counter()
    for q in letters:
        t0 = time.perf_counter()
        d.scan(q)
        out.append((time.perf_counter() - t0) * 1e3)
    total = time.perf_counter() - t_all
    print("%-12s total %.3f s  first eight: %s" % (mode, total, " ".join("%.1f" % x for x in out[:8])), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        for rep in range(3):
            for mode in ("none", "busy300", "busy1000", "sleep1000", "scan0twice"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", mode])
