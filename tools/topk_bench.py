#!/usr/bin/env python3
"""sw_topk: radix select against the full radix sort, N scores with Swiss-Prot-like ties, k = 10 and 1000."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch

from cudasw4_amd import capi


def main():
    ctx = capi.Context(0)
    for n in (1_000_000, 10_000_000, 100_000_000):
        s = torch.randint(0, 600, (n,), device="cuda").float()
        ids = torch.arange(n, dtype=torch.int32, device="cuda")
        for k in (10, 1000):
            row = []
            for path in ("sort", "select"):
                os.environ["CUDASW4_AMD_TOPK"] = path
                tb = capi.topk_temp_bytes(n, k)
                temp = torch.empty(tb, dtype=torch.uint8, device="cuda")
                os_ = torch.empty(k, dtype=torch.float32, device="cuda")
                oi = torch.empty(k, dtype=torch.int32, device="cuda")
                for rep in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ctx.topk(s.data_ptr(), ids.data_ptr(), n, k, os_.data_ptr(), oi.data_ptr(), temp.data_ptr(), tb, 0)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                row.append((path, dt * 1e3, oi[:3].tolist()))
                del temp
            print("n=%d k=%d: %s" % (n, k, ", ".join("%s %.3f ms %s" % r for r in row)), flush=True)
        del s, ids
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
