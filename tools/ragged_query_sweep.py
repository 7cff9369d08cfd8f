#!/usr/bin/env python3
"""Short queries on RAGGED subjects (the Swiss-Prot-like DB) through the C++ host driver: 8-lane vs 16-lane groups, per
query length and kernel configuration.  The peak DB's identical subjects hide LDS bank conflicts between the groups of a
wave (every group reads the same letter row), so the 8-lane limits (sw_api.hip: SW_LANES8_MAX_QUERY_*) are tuned here.
CUDASW4_AMD_LANES8_MAX_Q is read when the context is created: 0 = never 8 lanes, 1000000 = whenever one stripe fits.
Reports the whole scan (best of 3, host clock of the driver) and the bulk DP launch alone (HIP events).
A/B builds of the kernel library: LD_LIBRARY_PATH=cudasw4_amd/lib_NAME (tools/build_variant.sh).

    python tools/ragged_query_sweep.py [--db-size 570000] [--configs dpx,half2,float] [--lengths 48,96,...]"""
import argparse, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from cudasw4_amd import driver, synthdb

ap = argparse.ArgumentParser()
ap.add_argument("--db-size", type=int, default=synthdb.SPROT_SEQUENCES)
ap.add_argument("--configs", default="dpx,half2,float")
ap.add_argument("--lengths", default="48,96,144,189,222,256,272,288,304,320,352,384")
ap.add_argument("--modes", default="0,1000000")
ap.add_argument("--var", default="CUDASW4_AMD_LANES8_MAX_Q", help="the switch the modes are values of (CUDASW4_AMD_LANES4_MAX_Q: 4-lane groups)")
args = ap.parse_args()
chars, offsets, lengths = synthdb.sprot_like(args.db_size)
residues = float(lengths.astype(np.int64).sum())
rng = np.random.default_rng(1)
alphabet = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
queries = [alphabet[rng.integers(0, 20, n)].tobytes() for n in (int(x) for x in args.lengths.split(","))]
CONFIGS = {"dpx": (1, 1, 2, 2), "half2": (0, 0, 3, 3), "float": (3, 0, 3, 3), "dpxs32": (2, 1, 2, 2)}
for cname in args.configs.split(","):
    res = {}
    for mode in args.modes.split(","):
        os.environ[args.var] = mode
        d = driver.Driver(devices=[0], num_top=10, kinds=CONFIGS[cname])
        d.db_from_arrays(chars, offsets, lengths)
        d.upload()
        d.scan(queries[0])
        whole, bulk, shapes, scores = [], [], [], []
        for q in queries:
            best = 1e9
            d.record_kernel_events(True)
            for _ in range(3):
                best = min(best, d.scan(q)["seconds"])
            d.record_kernel_events(False)
            ev = d.take_kernel_events()
            main = [e for e in ev if e["subjects"] == max(x["subjects"] for x in ev)]
            whole.append(round(len(q) * residues / 1e9 / best))
            bulk.append(round(min(e["ms"] for e in main), 3))
            shapes.append("R%dx%d" % (main[0]["rows"], main[0]["lanes"]))
            scores.append(d.last_scores(0)[0].copy())
        res[mode] = (whole, bulk, shapes, scores)
        d.close()
    modes = args.modes.split(",")
    print(cname, "query lengths:", [len(q) for q in queries])
    for m in modes:
        print(cname, "%s=%-8s scan GCUPS:" % (args.var[12:], m), res[m][0])
        print(cname, "%s=%-8s bulk launch ms:" % (args.var[12:], m), res[m][1], res[m][2])
    if len(modes) == 2:
        a, b = res[modes[0]], res[modes[1]]
        print(cname, "same scores:", all((x == y).all() for x, y in zip(a[3], b[3])),
              "gain of the second mode %:", [round(100.0 * (y / x - 1), 1) for x, y in zip(a[0], b[0])])
