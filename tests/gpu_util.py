"""Shared helpers of the GPU parity tests (import only inside @pytest.mark.gpu tests)."""
import numpy as np

import oracle_lib as O


def gpu_modules():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from cudasw4_amd import capi, search
    return torch, capi, search


def scan_all_scores(search, capi, db_arrays, query, kernel_types=None, gop=-11, gex=-1, matrix=None,
                    merge=True, num_top=0, searcher=None):
    """Scores of every subject (original order) computed through the C ABI."""
    chars, offsets, lengths = db_arrays
    sc, so, sl, order = search.sort_db_by_length(chars, offsets, lengths)
    db = search.DeviceDB.from_arrays(sc, so, sl, device=0)
    s = searcher or search.Searcher(device=0, num_top=num_top, matrix=matrix if matrix is not None else O.blosum21(62),
                                    kernel_types=kernel_types, gop=gop, gex=gex, merge_partitions=merge)
    s.set_database(db)
    res = s.scan(query)
    sorted_scores = s.all_scores()
    out = np.empty_like(sorted_scores)
    out[order] = sorted_scores
    return out, res, order


def kinds_configs(search, capi):
    K = search.KernelTypeConfig
    return {
        "half2+float": K(capi.KIND_F16X2, capi.KIND_F16X2, capi.KIND_F32, capi.KIND_F32),
        "dpxs16+dpxs32": K.dpx(),
        "dpxs32": K(capi.KIND_I32, capi.KIND_I16X2, capi.KIND_I32, capi.KIND_I32),
        "float": K(capi.KIND_F32, capi.KIND_F16X2, capi.KIND_F32, capi.KIND_F32),
    }


def relatives(rng, q, n, lo, hi):
    """subjects that contain mutated copies of the query (substitutions, insertions, deletions) inside random flanks"""
    import numpy as np
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        s = rng.integers(0, 20, L).astype(np.int8)
        copy = []
        for c in q:
            r = rng.random()
            if r < 0.04:
                continue                                   # deletion
            if r < 0.08:
                copy.extend(rng.integers(0, 20, int(rng.integers(1, 12))).tolist())   # insertion
            copy.append(int(rng.integers(0, 20)) if r > 0.85 else int(c))
        copy = np.array(copy[:L], dtype=np.int8)
        at = int(rng.integers(0, L - len(copy) + 1))
        s[at:at + len(copy)] = copy
        out.append(s)
    return out
