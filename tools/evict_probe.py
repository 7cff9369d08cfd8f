#!/usr/bin/env python3
"""Which set-up step of the host driver leaves a 70-80 ms GPU stall behind in the first second (cold `align` runs show one
or two among their first queries; a 1 s sleep after the DB upload removes them)?  After the step under test a 20 us
kernel is launched and synchronised in a loop for 1.5 s; prints the long iterations and when they happened.  Fresh process
per mode.

  python tools/evict_probe.py            # parent
  python tools/evict_probe.py child MODE"""
import os
import subprocess
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def child(mode):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    dev = torch.device("cuda", 0)
    x = torch.zeros(1 << 16, device=dev)
    x.add_(1.0)
    torch.cuda.synchronize()
    keep = []
    if mode == "ctx":
        from cudasw4_amd import capi
        keep.append(capi.Context(0))
    elif mode == "driver":
        from cudasw4_amd import driver
        keep.append(driver.Driver(devices=[0], num_top=0, kinds=(0, 0, 3, 3)))
    elif mode in ("driver_db", "driver_db_upload"):
        from cudasw4_amd import driver
        d = driver.Driver(devices=[0], num_top=0, kinds=(0, 0, 3, 3))
        d.pseudo_db(1_000_000, 512)
        if mode == "driver_db_upload":
            d.upload()
        keep.append(d)
    elif mode.startswith("pseudo"):   # pseudo<N>: driver + pseudo DB of N subjects x 512, no upload
        from cudasw4_amd import driver
        d = driver.Driver(devices=[0], num_top=0, kinds=(0, 0, 3, 3))
        d.pseudo_db(int(mode[6:]), 512)
        keep.append(d)
    elif mode == "hostvec":           # only the host side of a big pseudo DB: numpy stand-in, no driver
        keep.append(np.full(512 << 20, 7, dtype=np.int8))
    elif mode == "malloc":
        keep.append(torch.empty(600 << 20, dtype=torch.int8, device=dev))
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    worst = []
    while time.perf_counter() - t_start < 1.5:
        t0 = time.perf_counter()
        x.add_(1.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if dt > 5:
            worst.append("%.0f ms at +%.0f ms" % (dt, (t0 - t_start) * 1e3))
    print("%-18s stalls: %s" % (mode, worst or "none"), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        for rep in range(3):
            for mode, env in (("none", {}), ("ctx", {}), ("driver", {}), ("pseudo1000", {}), ("driver_db", {}), ("driver_db_upload", {}),
                              ("driver_db", {"OMP_NUM_THREADS": "1"})):
                print(env or "", end=" ", flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", mode], env=dict(os.environ, **env))
