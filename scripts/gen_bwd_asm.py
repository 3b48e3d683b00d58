#!/usr/bin/env python3
"""Writes evcouplings_amd/csrc/plm_bwd_asm.inc: the K step of k_bwd_w (plm_kernels.hip) as ONE block of gfx950
assembly with a fixed register map.

Why assembly: a wave of k_bwd_w owns a 7 x 9 tile of int32 accumulator fragments (252 registers).  hipcc keeps such a
tile in the accumulation half of the register file only with copies through arch VGPRs around every MFMA (measured:
370-430 v_accvgpr moves per K step, or 600+ spilled registers, for every C++ spelling tried -- DESIGN.md 4.4).  Here
the accumulators ARE a[0:251], named in the instructions; the compiler only sees them in the clobber list.

Register map of a wave (arch VGPRs; everything below V_LO belongs to the compiler):
    a[0:251]      accumulators, fragment (f, c) at a[(f * FNW + c) * 4 ..+3]
    AF_A, AF_B    one-hot A fragments of the two 64-sequence halves of a K step (7 x 4 registers each); AF_A for step
                  s + 1 is produced during the second half of step s (it lives across the C++ code between two blocks:
                  the block's clobber list keeps the compiler's own values out of these registers, and the build checks
                  the generated ISA for any use of them outside the blocks -- scripts/check_bwd_asm.py)
    XA1, XN0      packed alignment bytes: second half of this step, first half of the next
    BF[0..2]      ring of three digit fragments (LDS -> register, two ahead of the MFMAs that consume them)
    T0, T1        temporaries of the one-hot expansion; VCNT the arrival counter as last read
LDS (set up by the kernel): rings of four digit tiles and four slots of alignment bytes, the arrival counter behind them.
The docstrings of step_block / dma_ops describe the schedule of a step and the synchronisation without barriers.
"""
import os

FM, FNW = 7, 9
V_LO = 176
AF_A, AF_B = 176, 204
XA1, XN0, T0, T1, VCNT = 232, 236, 240, 241, 242
BF = (244, 248, 252)          # ring of three digit fragments: the LDS read runs two fragments ahead of its MFMAs
V_HI = 255
NOVALU = int(os.environ.get("BWDW_NOVALU", "0"))    # timing experiments only (wrong results)
NOLDS = int(os.environ.get("BWDW_NOLDS", "0"))
NODMA = int(os.environ.get("BWDW_NODMA", "0"))
NOBAR = int(os.environ.get("BWDW_NOBAR", "0"))
NOCHECK = int(os.environ.get("BWDW_NOCHECK", "0"))
NACC = FM * FNW * 4


def vr(b, n=4):
    return f"v[{b}:{b + n - 1}]"


def acc(f, c):
    b = (f * FNW + c) * 4
    return f"a[{b}:{b + 3}]"


def expand(state, xa, af, lines_out):
    """the 4 x 2 VALU operations that turn 16 packed states into the one-hot fragment of `state`; returned as 4 pairs"""
    pairs = []
    for j in range(4):
        t = T0 if j % 2 == 0 else T1
        sb = "%[b0x]" if state == 0 else "%[st]"
        pairs.append([f"v_xad_u32 v{t}, v{xa + j}, {sb}, %[k7f]",
                      f"v_bfi_b32 v{af + 4 * state + j}, v{t}, 0, %[k80]"])
    return pairs


NPIECE = 5                    # digit-tile pieces of a wave per K step (waves with 4 repeat one), + 2 alignment pieces
NVMEM = NPIECE + 2
DMA_FRAGS = tuple(int(v) for v in os.environ.get("BWDW_DMAF", "8,9,10,11,12,13,14").split(","))   # fragments whose gaps 4 / 5 carry one LDS-DMA instruction (behind the check)
CNT_READ_FRAG, CNT_CHECK_FRAG = 5, 7     # the arrival counter is read at gap 4 of the first, tested in the second


def dma_ops():
    """(set M0, copy) pairs of one wave and K step: the tile and the alignment bytes of step s + 3.
    All 7 are always issued (sources clamped by the caller), so that `vmcnt(7)` at the end of a step means: everything
    issued in earlier steps has landed."""
    ops = []
    for k in range(NPIECE):
        if k == 0:
            m0 = "s_mov_b32 m0, %[m0t]"
        elif k < NPIECE - 1:
            m0 = "s_add_u32 m0, m0, 0x1000"
        else:
            m0 = "s_add_u32 m0, %[m0t], %[d4]"     # the piece a four-piece wave repeats
        ops.append((m0, f"global_load_lds_dwordx4 %[vo{k}], %[tsrc]"))
    ops.append(("s_mov_b32 m0, %[m0a]", "global_load_lds_dwordx4 %[acol], %[asrc]"))
    # (no offset: field here -- the instruction offset of an LDS-DMA load moves the LDS address as well)
    ops.append(("s_add_u32 m0, m0, 0x400", "global_load_lds_dwordx4 %[acol1], %[asrc]"))
    return ops


def frag_off(F):
    return (F % FNW) * 2048 + (F // FNW) * 1024


def step_block():
    """One K step s of a wave; on entry XA1, BF[0], BF[1] are in flight (read by the previous block / the prime block).
    A fragment is 7 MFMAs (gaps 0..6 behind them); its fillers are spread over the gaps, at most two per gap (a
    16-cycle MFMA leaves about three issue slots):
        gaps 0-3  the four xad / bfi pairs of the one-hot expansion (fragments of the first 7 columns of a half)
        gap 4     the LDS read of the fragment two ahead; M0 of this fragment's LDS-DMA instruction, if it has one
        gap 5     the state constant of the next fragment's expansion; the LDS-DMA instruction
        gap 6     the wait for the next fragment's digits
    The last two fragments read the first two fragments of step s + 1 (its tile landed a step ago), so the next block
    starts without an exposed LDS latency.
    No s_barrier in the loop (measured: 0.20 of 3.57 ms -- with one wave per SIMD nothing runs while a wave waits): a
    wave ARRIVES at the end of a step (vmcnt(7): its copies issued before this step have landed; ds_add on an LDS
    counter) and CHECKS in the middle of the next step that all four have arrived, before it issues the copies that
    overwrite the slot of the step before and before it reads the next tile: half a step of skew costs nothing."""
    L = []
    NF = 2 * FNW
    dma = dma_ops() if not NODMA else []
    L.append(f"ds_read_b128 {vr(XN0)}, %[lan]")
    # XA1 and fragment 0; fragment 1, the arrival add of the previous block and XN0 may still be in flight
    L.append("s_waitcnt lgkmcnt(%d)" % (2 if NOBAR else 3))
    for F in range(NF):
        C, H = F % FNW, F // FNW
        me = BF[F % 3]
        af = AF_A if H == 0 else AF_B
        pairs = []
        if C < FM and not NOVALU:
            pairs = expand(C, XA1 if H == 0 else XN0, AF_B if H == 0 else AF_A, L)
        gaps = [[] for _ in range(FM)]
        for j, pr in enumerate(pairs):
            gaps[j].extend(pr)
        if F == CNT_READ_FRAG and not NOBAR:
            # in front of this fragment's digit read: the counted wait of gap 6 covers it (LDS returns in order)
            gaps[4].append(f"ds_read_b32 v{VCNT}, %[cnt]")
        if F == CNT_CHECK_FRAG and not NOBAR and not NOCHECK:
            # every wave has finished step s - 1 (its reads of the slot this step's copies overwrite, its own copies of
            # step s + 2 landed)?  Normally yes at the first look; otherwise poll.
            gaps[0] += [f"v_readfirstlane_b32 %[st], v{VCNT}", "s_cmp_ge_u32 %[st], %[tgt]"]
            # (the poll is bounded: a workgroup that lost a wave traps instead of spinning forever)
            gaps[1] += ["s_cbranch_scc1 .Lbwdw_go_%=", "s_mov_b32 %[sp], 0x400000", ".Lbwdw_poll_%=:",
                        f"ds_read_b32 v{VCNT}, %[cnt]", "s_waitcnt lgkmcnt(0)", f"v_readfirstlane_b32 %[st], v{VCNT}",
                        "s_cmp_ge_u32 %[st], %[tgt]", "s_cbranch_scc1 .Lbwdw_go_%=", "s_sub_u32 %[sp], %[sp], 1",
                        "s_cmp_lg_u32 %[sp], 0", "s_cbranch_scc1 .Lbwdw_poll_%=", "s_trap 2", ".Lbwdw_go_%=:"]
        if not NOLDS:
            if F + 2 < NF:
                gaps[4].append(f"ds_read_b128 {vr(BF[(F + 2) % 3])}, %[lb] offset:{frag_off(F + 2)}")
            else:                                # fragments 0 / 1 of the next step, into the ring positions they will have
                gaps[4].append(f"ds_read_b128 {vr(BF[F + 2 - NF])}, %[lbn] offset:{frag_off(F + 2 - NF)}")
        if F in DMA_FRAGS and dma:
            m0, ld = dma[DMA_FRAGS.index(F)]
            gaps[4].append(m0)
            gaps[5].append(ld)
        C1 = (F + 1) % FNW
        if F + 1 < NF and 0 < C1 < FM and not NOVALU:
            gaps[5].append(f"s_add_u32 %[st], %[b0x], 0x{0x01010101 * C1:08x}")
        if F + 1 < NF and not NOLDS:
            gaps[6].append("s_waitcnt lgkmcnt(1)")      # all but the newest read: fragment F + 1 (and XN0) are in
        if F == NF - 1:
            # the next step's XA1 goes out before its first two fragments' reads are waited for (by the next block)
            gaps[0].append(f"ds_read_b128 {vr(XA1)}, %[lan] offset:1024")
        for f in range(FM):
            L.append(f"v_mfma_i32_16x16x64_i8 {acc(f, C)}, {vr(af + 4 * f)}, {vr(me)}, {acc(f, C)}")
            L.extend(gaps[f])
    if not NOBAR:
        # arrival: my copies issued before this step have landed, my LDS reads of this step are ahead of the add in the
        # LDS queue.  One lane adds (64 lanes on one address serialise in the LDS and hold up the reads queued behind
        # them: measured +0.4 ms); the kernel runs with full waves, so EXEC goes back to all ones.
        L.append(f"s_waitcnt vmcnt({NVMEM if dma else 0})")
        L += ["s_mov_b64 exec, 1", "ds_add_u32 %[cnt], %[one]", "s_mov_b64 exec, -1"]
    return L


def prime_block():
    """before the first step: AF_A from the first half of step k0's alignment bytes; XA1, BF[0], BF[1] of step k0"""
    L = [f"ds_read_b128 {vr(XN0)}, %[lan]", "s_waitcnt lgkmcnt(0)"]
    for C in range(FM):
        if C > 0:
            L.append(f"s_add_u32 %[st], %[b0x], 0x{0x01010101 * C:08x}")
        for p in expand(C, XN0, AF_A, L):
            L.extend(p)
    L.append(f"ds_read_b128 {vr(XA1)}, %[lan] offset:1024")
    L.append(f"ds_read_b128 {vr(BF[0])}, %[lbn]")
    L.append(f"ds_read_b128 {vr(BF[1])}, %[lbn] offset:{frag_off(1)}")
    return L


def issue_block():
    """the 7 LDS-DMA instructions of a step on their own (prologue: steps k0 and k0 + 1)"""
    L = []
    for m0, ld in dma_ops():
        L += [m0, "s_nop 0", ld]
    return L


def zero_block():
    return [f"v_accvgpr_write_b32 a{i}, 0" for i in range(NACC)]


def cstr(lines):
    return "\n".join(f'    "{l}\\n\\t"' for l in lines)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.environ.get("BWDW_OUT") or os.path.join(here, "..", "evcouplings_amd", "csrc", "plm_bwd_asm.inc")
    clob = ", ".join([f'"v{i}"' for i in range(V_LO, V_HI + 1)] + [f'"a{i}"' for i in range(NACC)])
    with open(out, "w") as fh:
        fh.write("// GENERATED by scripts/gen_bwd_asm.py -- do not edit; the register map is described there.\n")
        fh.write(f"#define PLM_BWDW_FM {FM}\n#define PLM_BWDW_FN {FNW}\n#define PLM_BWDW_VLO {V_LO}\n")
        fh.write(f"#define PLM_BWDW_CLOBBERS {clob}, \"m0\", \"scc\", \"memory\"\n")
        fh.write("#define PLM_BWDW_ZERO_ASM \\\n" + cstr(zero_block()).replace("\n", " \\\n") + "\n")
        fh.write(f"#define PLM_BWDW_NVMEM {NVMEM}\n")
        fh.write("#define PLM_BWDW_ISSUE_ASM \\\n" + cstr(issue_block()).replace("\n", " \\\n") + "\n")
        fh.write("#define PLM_BWDW_PRIME_ASM \\\n" + cstr(prime_block()).replace("\n", " \\\n") + "\n")
        fh.write("#define PLM_BWDW_STEP_ASM \\\n" + cstr(step_block()).replace("\n", " \\\n") + "\n")
    n = len(step_block())
    print(f"wrote {os.path.normpath(out)}: step block {n} instructions, "
          f"{2 * FNW * FM} MFMAs, {2 * FM * 8} expansion VALU")


if __name__ == "__main__":
    main()
