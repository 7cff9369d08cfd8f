#!/usr/bin/env python3
"""Spills that matter: scratch loads / stores INSIDE the innermost (quad) loops of the scan kernels of a gfx950 assembly
file (hipcc -S --cuda-device-only), next to the VALU / LDS / VMEM counts of those loops.  The code-object's
vgpr_spill_count also counts the set-up code a workgroup runs once.
    python tools/loop_spills.py file.s"""
import re
import sys


def kernels(path):
    cur, name = [], None
    for l in open(path):
        m = re.match(r"^(_ZN3swk(?:14sw_scan_kernel|21sw_scan_stream_kernel)\S+):", l)
        if m:
            if name:
                yield name, cur
            name, cur = m.group(1), []
        elif name:
            cur.append(l.rstrip("\n"))
            if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
                yield name, cur
                name, cur = None, []


def loops(lines):
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def main():
    for name, lines in kernels(sys.argv[1]):
        m = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)E", name) if "stream" not in name else None
        tag = "kind %s R %s lanes %s multi %s offs %s" % m.groups() if m else name
        inner = []
        ls = loops(lines)
        for a, b in ls:
            if any(a <= c and d <= b and (c, d) != (a, b) for c, d in ls):
                continue  # has a loop inside
            body = [x.strip() for x in lines[a:b + 1] if x.startswith("\t") and not x.strip().startswith(";")]
            if len(body) < 200:
                continue
            cnt = lambda p: sum(1 for x in body if x.startswith(p))
            inner.append("[%d instr: valu %d, ds %d, vmem %d, scratch ld %d st %d]" % (
                len(body), cnt("v_"), cnt("ds_"), cnt("global_"), cnt("scratch_load"), cnt("scratch_store")))
        print(tag, " ".join(inner))


if __name__ == "__main__":
    main()
