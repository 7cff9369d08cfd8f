"""GPU: a short pass of the randomised soak test (tests/fuzz_gpu.py): random DB shapes, queries around every stripe /
group-shape boundary, gap scores, matrices incl. the 25-letter tables, kernel-type configurations, both host drivers,
resident and streamed, int32 native and in fp32 lanes — every score and every top-10 list against the CPU oracle.
FUZZ_SECONDS / FUZZ_SEED lengthen or reseed it (the long runs are recorded in profiles/r02_results.md)."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_soak_short(monkeypatch):
    import fuzz_gpu
    for var in ("CUDASW4_AMD_I32_NATIVE", "CUDASW4_AMD_LANES8_MAX_Q", "CUDASW4_AMD_STREAM", "CUDASW4_AMD_GRID_CAP"):
        monkeypatch.setenv(var, os.environ.get(var, "-1" if "LANES8" in var else "0"))  # restored after the test
    seconds = os.environ.get("FUZZ_SECONDS", "25")
    seed = os.environ.get("FUZZ_SEED", "314")
    cases = fuzz_gpu.main(["--seconds", seconds, "--seed", seed, "--driver-bias", "0.3"])
    assert cases >= 20
