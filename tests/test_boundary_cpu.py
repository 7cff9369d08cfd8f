"""CPU: the reference-side binding documented in INTEGRATION.md §2 compiles against the reference's own headers and
include/cudasw4_amd.h, links against libcudasw4_amd.so, and follows the error convention (no GPU here: the first call
fails with SW_ERR_NO_DEVICE and the reference's CUERR-style check() exits 1)."""
import os
import re
import subprocess

import pytest

import oracle_lib as O

ROOT = O.ROOT
REF = "/root/reference/src"


def patched_source(tmp_path):
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 2. Patch a reference maintainer would apply"):]
    block = re.search(r"```cpp\n(.*?)```", sec, re.S).group(1)
    assert "sw_scan_batch(" in block and "sw_batch_create(" in block and "sw_topk(" in block and "sw_set_matrix(" in block
    harness = open(os.path.join(ROOT, "tests", "boundary", "binding_harness.cpp")).read()
    assert harness.count("//@@INTEGRATION_MD_PATCH@@") == 1
    src = str(tmp_path / "binding.cpp")
    open(src, "w").write(harness.replace("//@@INTEGRATION_MD_PATCH@@", block))
    return src


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "types.hpp")), reason="reference tree not present (GPU box)")
def test_documented_binding_compiles_links_and_follows_the_error_convention(tmp_path):
    src = patched_source(tmp_path)
    inc = ["-I" + REF, "-I" + os.path.join(ROOT, "include")]
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-variable",
                           "-Wno-unused-but-set-variable", "-Wno-comment", "-Wno-unknown-pragmas"] + inc + [src])
    libdir = os.path.join(ROOT, "cudasw4_amd", "lib")
    exe = str(tmp_path / "binding")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w"] + inc + [src, "-L" + libdir, "-lcudasw4_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run-time half of this test expects the no-device error")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 1 and "cudasw4_amd error -6" in p.stderr and "no HIP device" in p.stderr


def test_header_documents_the_max_subject_len_contract():
    h = open(os.path.join(ROOT, "include", "cudasw4_amd.h")).read()
    assert "max_subject_len" in h and "under-report" in h
