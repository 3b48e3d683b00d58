#!/usr/bin/env python3
"""Checks the ISA hipcc generated for k_bwd_w: outside the assembly blocks of plm_bwd_asm.inc (between ;;#ASMSTART / ;;#ASMEND)
no instruction may touch the registers those blocks keep live across the C++ code -- v[PLM_BWDW_VLO..255] and any
AccVGPR other than through the read-out helper.  Usage: check_bwd_asm.py <device .s file>"""
import re
import sys

V_LO = 176


def regs(line):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


def main(path):
    text = open(path).read()
    m = re.search(r"^_Z7k_bwd_wILi21EE[^\n]*\n(.*?)\n\s*\.section", text, re.S | re.M)
    if not m:
        sys.exit("k_bwd_w<21> not found in " + path)
    in_app, bad, n_app, n_mfma = False, [], 0, 0
    for line in m.group(1).split("\n"):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_app = True
            n_app += 1
            continue
        if t.startswith(";;#ASMEND"):
            in_app = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if in_app:
            n_mfma += t.startswith("v_mfma")
            continue
        if any(r >= V_LO for r in regs(t)) or re.search(r"\ba\[?\d", t) or t.startswith("v_mfma"):
            bad.append(t)
    if bad:
        print("compiler-generated code touches the registers of the assembly blocks:")
        for b in bad[:20]:
            print("   ", b)
        sys.exit(1)
    print(f"k_bwd_w<21>: {n_app} assembly blocks, {n_mfma} MFMAs inside them, no outside use of v{V_LO}+ / a*")


if __name__ == "__main__":
    main(sys.argv[1])
