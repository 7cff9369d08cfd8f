#!/usr/bin/env python3
"""Randomised soak test on the GPU box: random queries / databases / gap scores / matrices / kernel configurations /
host-driver modes, every score of every subject compared with the CPU oracle.  TEST INFRASTRUCTURE (the oracle is the
checker); tests/test_gpu_fuzz.py runs a short pass of it inside the GPU suite.

    python tests/fuzz_gpu.py --seconds 600 [--seed 1] [--driver-bias 0.3]

Prints one line per case and stops at the first mismatch with everything needed to reproduce it."""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import oracle_lib as O
from cudasw4_amd import capi, driver, search

LETTERS21 = b"ARNDCQEGHILKMFPSTWYVX"
LETTERS25 = b"ARNDCQEGHILKMFPSTWYVBJZX*"


def random_lengths(rng, n):
    mode = rng.integers(0, 5) if n >= 16 else 1
    if mode == 0:
        l = rng.integers(1, 60, n)
    elif mode == 1:
        l = rng.integers(1, 700, n)
    elif mode == 2:
        l = np.concatenate([rng.integers(1, 400, n - n // 8), rng.integers(1281, 5000, n // 8)])
    elif mode == 3:
        l = np.concatenate([rng.integers(100, 1280, n - 3), rng.integers(8001, 12000, 3)])
    else:
        l = np.maximum(1, np.full(n, int(rng.integers(1, 900))) - rng.integers(0, 2) * rng.integers(0, 8, n))  # equal or nearly equal
    return np.sort(l).astype(np.int32)


def mutate(rng, seq, rate):
    s = seq.copy()
    mask = rng.random(len(s)) < rate
    s[mask] = rng.integers(0, 20, int(mask.sum()))
    return s


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--driver-bias", type=float, default=0.0, help="fraction of cases forced to the default scoring so that they can go through the C++ driver")
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    env_before = {k: v for k, v in os.environ.items() if k.startswith("CUDASW4_AMD_")}
    t_end = time.time() + args.seconds
    case = 0
    K = search.KernelTypeConfig
    kind_cfgs = [(0, 0, 3, 3), (1, 1, 2, 2), (2, 1, 2, 2), (3, 0, 3, 3), (0, 1, 2, 3), (1, 0, 3, 2)]
    qlen_choices = [1, 2, 7, 8, 9, 15, 16, 17, 63, 127, 128, 129, 143, 240, 241, 288, 289, 383, 384, 385, 511, 767, 768, 769,
                    1023, 1535, 1536, 1537, 2047, 3071, 3073]
    while time.time() < t_end:
        case += 1
        seed = int(rng.integers(0, 2**31))
        r = np.random.default_rng(seed)
        n = int(r.choice([1, 2, 3, 31, 32, 33, 64, 65, 200, 700, 2000]))
        lengths = random_lengths(r, n)
        full25 = bool(r.integers(0, 6) == 0)
        which = int(r.choice([45, 50, 62, 80]))
        gop, gex = [(-11, -1), (-11, -1), (-13, -2), (-10, -1), (-5, -5), (-20, -3), (-1, -1), (-40, -12), (-3, -12), (-100, -30)][int(r.integers(0, 10))]
        if r.random() < args.driver_bias:
            full25, which, gop, gex = False, 62, -11, -1
        qlen = int(r.choice(qlen_choices)) if r.integers(0, 2) else int(r.integers(1, 5600))
        if qlen * int(lengths.astype(np.int64).sum()) > 3e10:
            qlen = max(1, int(3e10 / max(1, int(lengths.astype(np.int64).sum()))))
        q = r.integers(0, 25 if full25 else 21, qlen).astype(np.int8)
        seqs = [r.integers(0, 21 if r.integers(0, 4) == 0 else 20, int(l)).astype(np.int8) for l in lengths]
        # homologs of the query among the subjects: large scores, overflow lists, re-score
        if r.integers(0, 2) and not full25:
            for _ in range(int(r.integers(1, 6))):
                i = int(r.integers(0, n))
                a = int(r.integers(0, max(1, qlen - 1)))
                piece = mutate(r, np.minimum(q[a:a + int(lengths[i])], 20), float(r.choice([0.0, 0.05, 0.3])))
                seqs[i][:len(piece)] = piece
        chars, offsets, lens = O.make_db(seqs)
        if full25:
            m = driver.matrix25(which)
            mo = np.ascontiguousarray(m.reshape(25, 25)[:, list(range(20)) + [23]])
        else:
            m = driver.matrix(which)
            mo = O.blosum21(which)
        expect = O.scan(q, chars, offsets, lens, m21=mo, gop=gop, gex=gex)
        kinds = kind_cfgs[int(r.integers(0, len(kind_cfgs)))]
        host = "capi" if (full25 or which != 62 or (gop, gex) != (-11, -1) or r.integers(0, 2)) else "driver"
        # round 5: the pipelined entry points straight through the C ABI — every subject of the case as a pipeline of one-wave
        # stages (any length: short ones are one stage), and a re-score list split between the pipelined and the claim launch
        if host == "capi" and gop <= gex and r.integers(0, 4) == 0:
            host = "pipe"
        desc = "case %d seed %d host %s n %d lens %d..%d qlen %d kinds %s mat %d%s gap %d/%d" % (
            case, seed, host, n, int(lengths[0]), int(lengths[-1]), qlen, kinds, which, "_25" if full25 else "", gop, gex)
        if host == "pipe":
            import torch
            cpl = int(r.choice([0, 4, 8, 16]))
            os.environ.pop("CUDASW4_AMD_PIPE_CPL", None)
            if cpl:
                os.environ["CUDASW4_AMD_PIPE_CPL"] = str(cpl)
            ctx = capi.Context(0)
            ctx.set_matrix(m)
            ctx.set_query(q)
            ddb = search.DeviceDB.from_arrays(chars, offsets, lens, device=0)
            maxlen = int(lens.max())
            tb = ctx.scan_rows_pipelined_temp_bytes(n, maxlen)
            if tb > (3 << 30):
                print("skip " + desc + " (pipelined: hand-off array of %d MB)" % (tb >> 20), flush=True)
                continue
            slot = int(r.choice([0, 128, 168, 256]))
            ctx.set_rows_pipeline_slot(slot)
            dsc = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
            did = torch.full((n,), -7, dtype=torch.int32, device="cuda")
            fails = torch.zeros(1, dtype=torch.int32, device="cuda")
            over = torch.zeros(2, dtype=torch.int32, device="cuda")
            limit = int(r.choice([60, 2048]))
            mode = int(r.integers(0, 2))
            if mode == 0:   # the whole range pipelined
                temp = torch.empty(max(tb, 256), dtype=torch.uint8, device="cuda")
                ctx.scan_rows_pipelined(ddb.chars.data_ptr(), ddb.offsets.data_ptr(), ddb.lengths.data_ptr(), 0, n, maxlen, gop, gex,
                                        dsc.data_ptr(), did.data_ptr(), 0, fails.data_ptr(), temp.data_ptr(), tb, 0,
                                        over.data_ptr(), over.data_ptr() + 4, limit)
            else:           # a re-score list of every subject: the long entries pipelined, the rest by the claim launch
                lst = torch.from_numpy(r.permutation(n).astype(np.int32)).cuda()
                cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
                minlen = int(r.choice([1, 64, 300, 1500]))
                tb1 = ctx.rescore_overflow_pipelined_temp_bytes(maxlen)
                kk = capi.KIND_F32 if r.integers(0, 2) else capi.KIND_I32
                tb2 = ctx.scan_temp_bytes(kk, -1, n, maxlen)
                temp = torch.empty(max(tb1, tb2, 256), dtype=torch.uint8, device="cuda")
                ctx.rescore_overflow_pipelined(lst.data_ptr(), cnt.data_ptr(), n, ddb.chars.data_ptr(), ddb.offsets.data_ptr(), ddb.lengths.data_ptr(),
                                               maxlen, minlen, gop, gex, dsc.data_ptr(), did.data_ptr(), 0, fails.data_ptr(), limit,
                                               over.data_ptr(), temp.data_ptr(), temp.numel())
                ctx.rescore_overflow_claim(kk, lst.data_ptr(), cnt.data_ptr(), n, ddb.chars.data_ptr(), ddb.offsets.data_ptr(), ddb.lengths.data_ptr(),
                                           maxlen, gop, gex, dsc.data_ptr(), did.data_ptr(), 0, temp.data_ptr(), temp.numel(), limit, over.data_ptr())
            torch.cuda.synchronize()
            got = dsc.cpu().numpy().astype(np.int64)
            desc += " cpl %d slot %d %s" % (cpl, slot, "range" if mode == 0 else "list")
            want_over = int((expect >= limit).sum())
            if int(fails.item()) != 0 or int(over[0].item()) != want_over or (did.cpu().numpy() != np.arange(n)).any():
                print("FAIL pipelined bookkeeping", int(fails.item()), over.cpu().numpy(), want_over, desc)
                sys.exit(1)
            top = tuple(x.tolist() for x in O.topk(expect, min(10, n)))   # (no top-K in this case: the scores are what is checked)
            ctx.close()
        elif host == "capi":
            kt = K(*kinds)
            os.environ["CUDASW4_AMD_I32_NATIVE"] = str(int(r.integers(0, 2)))
            os.environ["CUDASW4_AMD_LANES8_MAX_Q"] = str(int(r.choice([-1, -1, 0, 100000])))
            os.environ["CUDASW4_AMD_LANES4_MAX_Q"] = str(int(r.choice([-1, 0, 100000, 100000])))   # (explicit: quads whatever the batch count)
            os.environ["CUDASW4_AMD_LANES4_MAX_SUBJECT"] = str(int(r.choice([-1, -1, 100000])))
            os.environ["CUDASW4_AMD_STREAM"] = str(int(r.choice([1, 16, 16, 3])))
            os.environ["CUDASW4_AMD_GRID_CAP"] = str(int(r.choice([0, 0, 1, 3])))   # few workgroups: long claims of the streamed kernels
            s = search.Searcher(device=0, num_top=min(10, n), matrix=m, kernel_types=kt, gop=gop, gex=gex,
                                merge_partitions=bool(r.integers(0, 2)))
            s.set_database(search.DeviceDB.from_arrays(chars, offsets, lens, device=0))
            res = s.scan(q)
            got = s.all_scores()
            top = (res.scores, res.reference_ids)
            del s
        else:
            letters = bytes(LETTERS21[c] for c in q)
            devs = [[0], [0, 0], [0, 0, 0]][int(r.integers(0, 3))]
            kw, mode = {}, "resident"
            pick = int(r.integers(0, 3))
            if pick == 1:
                kw, mode = dict(max_gpu_mem=1, max_batch_bytes=int(r.choice([2000, 20000, 300000]))), "streamed"
            elif pick == 2:
                # hybrid residency: a limit that leaves room for a random part of a shard next to the staging buffers
                batch = int(r.choice([2000, 20000, 300000]))
                shard = max(1, int(offsets[-1]) // len(devs))
                kw = dict(max_gpu_mem=int((float(r.uniform(0.1, 0.9)) * shard + 3 * (batch + 64) + 64) / 0.75) + 24 * n + 8,
                          max_batch_bytes=batch)
                mode = "hybrid"
            # round 4: long subjects as windows (forced on wherever the span bound cuts a subject, off, or left to the
            # driver's estimate) and the re-score service (forced on / off / feedback; it only runs for a driver that is
            # alone on its device)
            wmode = int(r.integers(0, 3))
            os.environ.pop("CUDASW4_AMD_WINDOWS", None)
            if wmode == 0:
                os.environ["CUDASW4_AMD_WINDOWS"] = "always"
            elif wmode == 1:
                os.environ["CUDASW4_AMD_WINDOWS"] = "0"
            smode = int(r.integers(0, 3))
            os.environ.pop("CUDASW4_AMD_RESCORE_SERVICE", None)
            if smode < 2:
                os.environ["CUDASW4_AMD_RESCORE_SERVICE"] = str(smode)
            mode += " windows=%s service=%s" % (["always", "off", "auto"][wmode], ["off", "on", "auto"][smode])
            os.environ.pop("CUDASW4_AMD_TAIL_OVERLAP", None)
            if r.integers(0, 4) == 0:
                os.environ["CUDASW4_AMD_TAIL_OVERLAP"] = "0"
            # which subjects run pipelined (never / every long one / by the estimate with an extreme share), the span width, the
            # pipelined re-score, the streamed packed kernels (round 6: one batch at a time / short / long claims)
            r5 = {"CUDASW4_AMD_PIPELINES": r.choice(["", "0", "always"]), "CUDASW4_AMD_PIPELINE_SHARE": r.choice(["", "0.02", "5"]),
                  "CUDASW4_AMD_PIPELINE_RESCORE_SHARE": r.choice(["", "0.001", "1000"]), "CUDASW4_AMD_PIPE_CPL": r.choice(["", "4", "8", "16"]),
                  "CUDASW4_AMD_STREAM": r.choice(["", "1", "4"]), "CUDASW4_AMD_GRID_CAP": r.choice(["", "2", "5"]),
                  "CUDASW4_AMD_SIDE_RESERVE": r.choice(["", "0", "64"]), "CUDASW4_AMD_LANES4_MAX_Q": r.choice(["", "0", "100000"]),
                  "CUDASW4_AMD_SPLIT34_MAX_LANES": r.choice(["", "0", "8"])}
            for k5, v5 in r5.items():
                os.environ.pop(k5, None)
                if v5:
                    os.environ[k5] = str(v5)
            mode += " r5=" + ",".join("%s=%s" % (k5.replace("CUDASW4_AMD_", ""), v5) for k5, v5 in r5.items() if v5)
            d = driver.Driver(devices=devs, num_top=min(10, n), kinds=kinds, **kw)
            d.db_from_arrays(chars, offsets, lens)
            if r.integers(0, 3) == 0:
                # two queries in flight (submit / collect): the last one's results are the ones checked.  On a resident
                # shard every query but the first runs on the GPU's other lane, gated on the dry signal of the one before
                # (tail hand-over; CUDASW4_AMD_TAIL_OVERLAP=0: one lane)
                others = [bytes(LETTERS21[c] for c in r.integers(0, 20, int(r.integers(1, 400)))) for _ in range(int(r.integers(1, 4)))]
                rr = d.scan_many(others + [letters])[-1]
                mode += " pipelined x%d lanes=%s gated=%d" % (len(others) + 1, os.environ.get("CUDASW4_AMD_TAIL_OVERLAP", "auto"), d.tail_overlaps())
            else:
                rr = d.scan(letters)
            ids, sc = d.all_scores()
            got = np.empty_like(sc)
            got[ids] = sc
            top = (rr["scores"], rr["ids"])
            # the reference's overflow statistic: subjects of the packed partitions whose exact score reaches the limit
            packed = np.array([(kinds[0] if l <= 1280 else kinds[1] if l <= 8000 else kinds[2]) for l in lens])
            limit = np.where(packed == 0, 2048, np.where(packed == 1, 25000, 2**30))
            want_ovf = int((expect >= limit).sum())
            if rr["num_overflows"] != want_ovf or rr["num_rescored"] < want_ovf:
                print("FAIL overflow statistic", rr["num_overflows"], rr["num_rescored"], want_ovf, desc)
                sys.exit(1)
            d.close()
            desc += " devs %d %s" % (len(devs), mode)
        es, ei = O.topk(expect, min(10, n))
        ok = (got == expect).all() and list(top[0]) == es.tolist() and list(top[1]) == ei.tolist()
        print(("ok   " if ok else "FAIL ") + desc, flush=True)
        if not ok:
            bad = np.nonzero(got != expect)[0]
            print("mismatches at", bad[:10], "got", got[bad[:10]], "expect", expect[bad[:10]], "lengths", lens[bad[:10]])
            print("top got", top, "expect", es, ei)
            sys.exit(1)
    # (tests/test_gpu_fuzz.py runs this in the pytest process: leave no switch behind for the tests after it)
    for k in list(os.environ):
        if k.startswith("CUDASW4_AMD_") and k not in env_before:
            os.environ.pop(k)
    for k, v in env_before.items():
        os.environ[k] = v
    print("fuzz: %d cases, no mismatch" % case)
    return case


if __name__ == "__main__":
    main()
