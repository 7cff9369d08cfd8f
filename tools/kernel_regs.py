#!/usr/bin/env python3
"""Register / spill / LDS figures of the compiled DP kernels, read from the code-object metadata of the
per-kind object files (no GPU needed):  python tools/kernel_regs.py [f16x2 i16x2 i32 f32] [--spills-only]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(kind):
    obj = os.path.join(ROOT, "cudasw4_amd", "lib", "obj", "sw_kind_%s.o" % kind)
    with tempfile.TemporaryDirectory() as td:
        co, fb = os.path.join(td, "k.co"), os.path.join(td, "k.fatbin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, obj],
                              stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out = []
    for block in notes.split("- .agpr_count")[1:]:
        g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, block)
        name = g("name").group(1)
        m = re.search(r"sw_scan_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)E", name)
        if not m:
            continue
        out.append({"kind": int(m.group(1)), "R": int(m.group(2)), "lanes": int(m.group(3)), "multi": int(m.group(4)),
                    "offs": int(m.group(5)), "vgpr": int(g("vgpr_count").group(1)), "spill": int(g("vgpr_spill_count").group(1)),
                    "sgpr_spill": int(g("sgpr_spill_count").group(1)), "lds": int(g("group_segment_fixed_size").group(1)),
                    "scratch": int(g("private_segment_fixed_size").group(1))})
    return sorted(out, key=lambda k: (k["lanes"], k["offs"], k["multi"], k["R"]))


if __name__ == "__main__":
    kinds = [a for a in sys.argv[1:] if not a.startswith("--")] or ["f16x2", "i16x2", "i32", "f32"]
    spills_only = "--spills-only" in sys.argv
    for kind in kinds:
        for k in kernels(kind):
            if spills_only and not (k["spill"] or k["scratch"]):
                continue
            print("%-6s R=%2d lanes=%2d multi=%d offs=%d  vgpr=%3d spill=%3d scratch=%4d lds=%6d" % (
                kind, k["R"], k["lanes"], k["multi"], k["offs"], k["vgpr"], k["spill"], k["scratch"], k["lds"]))
