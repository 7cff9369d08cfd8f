#!/usr/bin/env python3
"""Short queries on the peak DB, 8-lane vs 16-lane groups (CUDASW4_AMD_LANES8_MAX_Q is read when the context is created)."""
import os, sys, time, json
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cudasw4_amd import capi, driver, search
_, letters = driver.read_sequences(os.path.join(ROOT, "tests", "golden", "allqueries.fasta"))
queries = [driver.encode(q) for q in letters]
rng = np.random.default_rng(1)
extra = [rng.integers(0, 20, n).astype(np.int8) for n in (48, 96, 280, 320, 384)]
golden = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_scores.json")))["pseudo"]
for L in (512, 128):
    db = search.DeviceDB.pseudo(1_000_000, L, driver.pseudo_sequence(L, 42), device=0)
    for kname, kind in (("half2", 0), ("dpxs32", 2)):
        big = capi.KIND_F32 if kind in (0, 3) else capi.KIND_I32
        small = kind if kind in (0, 1) else 1
        kt = search.KernelTypeConfig(kind, small, big, big)
        res = {}
        for mode in ("0", "1000000"):
            os.environ["CUDASW4_AMD_LANES8_MAX_Q"] = mode
            s = search.Searcher(device=0, num_top=0, matrix=driver.matrix(62), kernel_types=kt)
            s.set_database(db)
            s.scan(queries[0])
            out = []
            for qi, q in list(enumerate(queries[:5])) + [(None, e) for e in extra]:
                r = s.scan(q)
                r = s.scan(q)
                sc = s.all_scores()
                ok = qi is None or (int(sc.min()) == int(sc.max()) == golden[str(L)][qi])
                out.append((len(q), round(r.gcups, 0), ok))
            res[mode] = out
            del s
        print(L, kname, "16-lane:", res["0"])
        print(L, kname, " 8-lane:", res["1000000"])
    del db
    torch.cuda.empty_cache()
