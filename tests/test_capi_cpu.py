"""CPU: the C-ABI library loads, exports every declared symbol and fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = O.ROOT


def _lib_path():
    return os.path.join(ROOT, "cudasw4_amd", "lib", "libcudasw4_amd.so")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib_path()):
        import __graft_entry__ as g
        g.build()
    return _lib_path()


def test_header_symbols_are_exported(built):
    # the boundary (cudasw4_amd.h) and the building blocks of its batch engine (cudasw4_amd_engine.h)
    boundary = set(re.findall(r"^[a-z_0-9 *]*\b(sw_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "cudasw4_amd.h")).read(), re.M))
    blocks = set(re.findall(r"^[a-z_0-9 *]*\b(sw_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "cudasw4_amd_engine.h")).read(), re.M))
    assert {"sw_scan_batch", "sw_scan_partition", "sw_set_query", "sw_topk"} <= boundary and not (boundary & blocks)
    assert len(boundary) <= 32 and "sw_set_start_signal" in blocks and "sw_rescore_service" in blocks   # (round 5: 45 exports in one header)
    declared = boundary | blocks
    from cudasw4_amd import capi
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = ctypes.CDLL(built)
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_plan_without_gpu(built):
    from cudasw4_amd import capi
    assert "gfx950" in capi.version()
    for kind in (capi.KIND_F16X2, capi.KIND_I16X2):
        for q in (1, 63, 64, 65, 144, 512, 513, 5478, 40000):
            r, s = capi.plan_query(kind, q)
            assert 1 <= r <= 48 and 16 * r * s >= q > 16 * r * s - 16 * s   # padding below one row per lane and stripe
            assert (q + 767) // 768 <= s <= max((q + 767) // 768 + 2, (q + 511) // 512 + 1)
    for kind, rmax in ((capi.KIND_I32, 48), (capi.KIND_F32, 36)):
        for q in (1, 256, 257, 567, 5478):
            r, s = capi.plan_query(kind, q)
            assert 1 <= r <= rmax and 16 * r * s >= q > 16 * r * s - 16 * s
    assert capi.plan_query(capi.KIND_I32, 657)[1] == 1 and capi.plan_query(capi.KIND_F32, 657)[1] == 2
    with pytest.raises(capi.SwError):
        capi.plan_query(7, 100)


def test_planner_prefers_three_wave_stripes(built, monkeypatch):
    """Packed multi-stripe kernels up to 32 rows per lane keep three waves per SIMD (sw_dp_kernel.hpp:
    SWK_WAVES3_MAX_R_MULTI), and a step-row of the taller two-wave kernels is priced 4.5 % higher: the planner takes
    32-row stripes unless the taller plan saves more than that in padding and per-step overhead (measured per query on the
    peak DB, profiles/r04_results.md).  CUDASW4_AMD_TWO_WAVE_PENALTY=1 is the planner of rounds 1-3."""
    from cudasw4_amd import capi
    monkeypatch.delenv("CUDASW4_AMD_TWO_WAVE_PENALTY", raising=False)
    want = {850: (27, 2), 1000: (32, 2), 1500: (47, 2), 2005: (32, 4), 2504: (32, 5), 3005: (47, 4), 3564: (32, 7),
            4061: (32, 8), 4548: (32, 9), 4743: (30, 10), 5147: (27, 12), 5478: (29, 12)}
    for kind in (capi.KIND_F16X2, capi.KIND_I16X2):
        for q, plan in want.items():
            assert capi.plan_query(kind, q) == plan, (q, capi.plan_query(kind, q))
        for q in (100, 512, 729, 768):      # one stripe whenever the query fits one
            assert capi.plan_query(kind, q)[1] == 1
    assert capi.plan_query(capi.KIND_I32, 5478) == (43, 8)      # the 32-bit kinds are not affected
    monkeypatch.setenv("CUDASW4_AMD_TWO_WAVE_PENALTY", "1")
    old = {850: (27, 2), 2005: (42, 3), 2504: (40, 4), 4061: (43, 6), 5478: (43, 8)}
    for q, plan in old.items():
        assert capi.plan_query(capi.KIND_F16X2, q) == plan


def test_no_cpu_fallback(built):
    """Without a GPU the product must fail loudly, never compute on the CPU."""
    import torch
    from cudasw4_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert capi.device_count() == 0
    with pytest.raises(capi.SwError) as ei:
        capi.Context(0)
    assert ei.value.code == -6


def test_partition_table_matches_reference():
    from cudasw4_amd import search
    g = O.golden("ref_tables.json")
    assert search.PARTITION_BOUNDARIES.tolist() == g["partition_boundaries"]


def test_kernel_type_validation():
    from cudasw4_amd import search, capi
    K = search.KernelTypeConfig
    K().validate()
    K.dpx().validate()
    with pytest.raises(ValueError):
        K(many_pass_small=capi.KIND_F32).validate()
    with pytest.raises(ValueError):
        K(many_pass_large=capi.KIND_F16X2).validate()
    with pytest.raises(ValueError):
        K(overflow=capi.KIND_I16X2).validate()
    k = K()
    assert k.kind_for_partition(0) == capi.KIND_F16X2 and k.kind_for_partition(35) == capi.KIND_F32


def test_sort_db_and_merge_topk():
    from cudasw4_amd import search
    rng = np.random.default_rng(3)
    seqs = [rng.integers(0, 20, int(l)).astype(np.int8) for l in (9, 3, 7, 3, 12)]
    chars, offsets, lengths = O.make_db(seqs)
    sc, so, sl, order = search.sort_db_by_length(chars, offsets, lengths)
    assert sl.tolist() == [3, 3, 7, 9, 12] and order.tolist() == [1, 3, 2, 0, 4]
    for k, i in enumerate(order):
        np.testing.assert_array_equal(sc[int(so[k]):int(so[k]) + int(sl[k])], seqs[i])
    assert all(int(o) % 4 == 0 for o in so)
    s, i = search.merge_topk([([9, 5, 5], [4, 1, 7]), ([9, 6], [2, 11])], 4)
    assert s.tolist() == [9, 9, 6, 5] and i.tolist() == [2, 4, 11, 1]


def test_full_25_letter_tables_and_query_encoder():
    """types.hpp:205-396: the host library's 25 x 25 tables equal the reference's; every entry of the X column is
    negative (subject code 20 and all padding are scored with it); the 25-letter query encoder keeps B, J, Z, X, *."""
    import numpy as np
    from cudasw4_amd import driver
    g = O.golden("ref_tables.json")
    for which in (45, 50, 62, 80):
        m = driver.matrix25(which).reshape(25, 25)
        assert m.reshape(-1).tolist() == g["blosum25"][str(which)]
        assert (m[:, 23] < 0).all() and (m[23, :] < 0).all()
        assert (m[:20, :20] == driver.matrix(which).reshape(21, 21)[:20, :20]).all()
        assert m[24, 24] == 1  # '*' vs '*' scores +1: why these tables need a padding row of their own
    assert driver.encode25("ARNDCQEGHILKMFPSTWYVBJZX*").tolist() == list(range(25))
    assert driver.encode25("abU-").tolist() == [23, 23, 23, 23]
    assert driver.encode("BJZX*").tolist() == [20] * 5
