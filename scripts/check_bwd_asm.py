#!/usr/bin/env python3
"""Checks the ISA hipcc generated for k_bwd_w and k_fwd_w: outside the assembly blocks of plm_bwd_asm.inc /
plm_fwd_asm.inc (between ;;#ASMSTART / ;;#ASMEND) no instruction may touch the registers those blocks keep live across
the C++ code -- v[V_LO..255] and any AccVGPR other than through the read-out helpers.
Usage: check_bwd_asm.py <device .s file>   (hipcc -S --cuda-device-only of plm_kernels.hip)"""
import re
import sys

KERNELS = (("k_bwd_w<21>", "_Z7k_bwd_wILi21EE", 176), ("k_fwd_w", "_Z7k_fwd_w7PlmDims7FwdArgs", 92))


def regs(line):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


def check(text, name, symbol, v_lo):
    m = re.search(r"^" + re.escape(symbol) + r"[^\n]*\n(.*?)\n\s*\.section", text, re.S | re.M)
    if not m:
        sys.exit(name + " not found")
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
            n_mfma += t.startswith("v_mfma") or t.startswith("v_smfmac")
            continue
        if any(r >= v_lo for r in regs(t)) or re.search(r"\ba\[?\d", t) or t.startswith("v_mfma") or t.startswith("v_smfmac"):
            bad.append(t)
    if bad:
        print(name + ": compiler-generated code touches the registers of the assembly blocks:")
        for b in bad[:20]:
            print("   ", b)
        return False
    print(f"{name}: {n_app} assembly blocks, {n_mfma} MFMAs inside them, no outside use of v{v_lo}+ / a*")
    return True


def main(path):
    text = open(path).read()
    ok = [check(text, *k) for k in KERNELS]
    sys.exit(0 if all(ok) else 1)


if __name__ == "__main__":
    main(sys.argv[1])
