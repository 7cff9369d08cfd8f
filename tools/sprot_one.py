#!/usr/bin/env python3
"""One query of the Swiss-Prot-like workload through one host driver (for rocprofv3 kernel traces): sprot_one.py cpp|py [qi]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cudasw4_amd import driver, search, synthdb
which = sys.argv[1]
qi = int(sys.argv[2]) if len(sys.argv) > 2 else 19
chars, offsets, lengths = synthdb.sprot_like()
_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
if which == "cpp":
    d = driver.Driver(devices=[0], num_top=10, kinds=(1, 1, 2, 2))
    d.db_from_arrays(chars, offsets, lengths)
    d.upload()
    for mode in (0, 1, 2, 0, 1):
        d.record_kernel_events(mode)
        secs = []
        for _ in range(6):
            r = d.scan(letters[qi])
            secs.append(round(r["seconds"] * 1e3, 2))
        ev = d.take_kernel_events()
        print("record mode", mode, "ms per scan", secs, [(e["part_id"], round(e["ms"], 1)) for e in ev[-3:]])
else:
    db = search.DeviceDB.from_arrays(chars, offsets, lengths, device=0)
    s = search.Searcher(device=0, num_top=10, matrix=driver.matrix(62), kernel_types=search.KernelTypeConfig.dpx())
    s.set_database(db)
    for _ in range(3):
        r = s.scan(driver.encode(letters[qi]))
    print(r.seconds, r.gcups)
