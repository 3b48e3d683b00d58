#!/usr/bin/env python3
"""Writes evcouplings_amd/csrc/plm_fwd_asm.inc: the K loop body of k_fwd_w (plm_kernels.hip) as blocks of gfx950
assembly with a fixed register map -- the forward GEMM with 512 sequences per workgroup.

Shape: workgroup = 4 waves (one per SIMD) x 128 sequences = 8 row fragments each, one 16-site block, ONE group of 7
states (three workgroups share a site block).  A wave's 8 x 7 accumulator fragments are a[0:223].  Against k_fwd (8
waves x 2 row fragments x 21 states) every B fragment read from LDS feeds 8 MFMAs instead of 2 -- k_fwd runs at the
LDS read peak (2 KB per two 16-cycle MFMAs and wave: 128 B/clk/CU) with its matrix cores 51 % busy -- and a tile
streamed from L2 serves 512 sequences instead of 256.

A block is one instruction SLICE ci of a 32-site block u: both planes (hi, lo) of the 7 states = 14 B fragments x 8 row
fragments = 112 v_smfmac_f32_16x16x64_f16.  With the K order of k_expand (slice = 4 neighbouring sites x one group of 4
states) the compressed one-hot fragment of a row fragment and slice is
    value of pair p  = 1.0 where (state - 1) >> 2 of site 4 (ci % 2) + p equals ci / 2      (byte compare, SDWA)
    2-bit positions  = (state - 1) & 3 of those four sites                                  (per u, not per slice)
and the fragments of slice ci + 1 are built behind the MFMAs of slice ci (9 VALU per row fragment), ONE operation per gap and
never in a gap that holds an LDS read or a copy (SCHED below: round 6, after scripts/ubench/mfma_coissue.hip).

Register map of a wave (arch VGPRs from V_LO; everything below belongs to the compiler):
    AS[set][m]   4 value dwords, AI[set][m] the index word; slice ci computes on set ci % 2 and builds the other
    G[m][d]      (state - 1) >> 2 of the 8 sites of this lane, one byte per site (two dwords)       } of the CURRENT u,
    IX[m][d]     the index word of dword d                                                          } set by UPREP
    XN[m]        the raw alignment bytes of the NEXT u (global_load_dwordx2, consumed by the next UPREP)
    BR[0..3]     ring of B fragments (8 registers: slots 0-7 | slots 8-15), read two fragments ahead
Synchronisation, LDS-DMA and the arrival counter work as in k_bwd_w (scripts/gen_bwd_asm.py): ring of four tiles in LDS,
copies of step s + 3 issued behind the check in step s, vmcnt(7) + arrival at the end, no s_barrier.
"""
import os

NM, QG = 8, 7                 # row fragments of a wave, states of a workgroup
NF = 2 * QG                   # B fragments of a slice (plane-major)
V_LO = 92
AS = (92, 124)                # two sets of NM x 4 value dwords (64-bit aligned tuples)
AI = (156, 164)               # ... and their NM index words
G, IX, XN = 172, 188, 204     # NM x 2 each
BR = (220, 228, 236, 244)
# ring entry of B fragment F: three consecutive fragments (cyclically: the last two fragments of a block read the first
# two of the next) must sit in different entries, and 14 is not a multiple of 3 -- four entries, period 14
RING = (0, 1, 2, 3, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2)
C1, VCNT, T0, T1 = 252, 253, 254, 255      # f16 1.0 constant, arrival counter, temporaries
NACC = NM * QG * 4
NVMEM = 7
CNT_READ_FRAG, CNT_CHECK_FRAG = 4, 6
DMA_FRAGS = (7, 8, 9, 10, 11, 12, 13)
SLOT = 2 * NF * 1024          # LDS bytes of a step's tile: [plane][half][state] x 1 KB
NOBUILD = int(os.environ.get("FWDW_NOBUILD", "0"))      # timing experiments only (wrong results)
NOLDS = int(os.environ.get("FWDW_NOLDS", "0"))
NODMA = int(os.environ.get("FWDW_NODMA", "0"))
NOSYNC = int(os.environ.get("FWDW_NOSYNC", "0"))
RD_GAPS = tuple(int(v) for v in os.environ.get("FWDW_RDGAPS", "5,6").split(","))
DMA_GAPS = tuple(int(v) for v in os.environ.get("FWDW_DMAGAPS", "3,4").split(","))
# Schedule of the fillers (round 6, scripts/ubench/mfma_coissue.hip): behind a v_smfmac of the one wave a SIMD hosts ONE
# VALU operation issues for free and so does one LDS read -- but a VALU operation and an LDS read (or an LDS-DMA copy) in
# the SAME gap cost 11 / 46 cycles.  SCHED 1 keeps them apart: the one-hot build is spread over the gaps 0-4 and 7 of
# fragments 0..12, gaps 5 / 6 belong to the LDS reads, gap 4 of the copying fragments to the copy.  SCHED 0 = round 4's
# (build in every gap of the first 8 fragments).
SCHED = int(os.environ.get("FWDW_SCHED", "1"))
VALU_GAPS = tuple(int(v) for v in os.environ.get("FWDW_VALUGAPS", "0,1,2,3,4,7").split(","))
CNT_GAP = int(os.environ.get("FWDW_CNTGAP", str(RD_GAPS[0])))      # gap of the arrival counter's read in fragment CNT_READ_FRAG


def vr(b, n):
    return f"v[{b}:{b + n - 1}]" if n > 1 else f"v{b}"


def acc(m, a):
    b = (m * QG + a) * 4
    return f"a[{b}:{b + 3}]"


def a_set(s, m):
    return AS[s] + 4 * m


def a_idx(s, m):
    return AI[s] + m


def build_ops(s, m, d):
    """the compressed fragment of row fragment m for a slice on dword d (0 / 1) into set s; %[kg] = its state group"""
    ops = []
    for p in range(4):
        ops.append(f"v_cmp_eq_u32_sdwa vcc, %[kg], v{G + 2 * m + d} src0_sel:DWORD src1_sel:BYTE_{p}")
        ops.append(f"v_cndmask_b32_e32 v{a_set(s, m) + p}, 0, v{C1}, vcc")
    ops.append(f"v_mov_b32_e32 v{a_idx(s, m)}, v{IX + 2 * m + d}")
    return ops


def b_reads(F, base):
    """the two halves of B fragment F of the slot at `base` into its ring entry"""
    plane, a = F // QG, F % QG
    r = BR[RING[F]]
    return [f"ds_read_b128 {vr(r, 4)}, {base} offset:{(plane * NF + a) * 1024}",
            f"ds_read_b128 {vr(r + 4, 4)}, {base} offset:{(plane * NF + QG + a) * 1024}"]


def dma_ops():
    ops = []
    for k in range(NVMEM):
        m0 = "s_mov_b32 m0, %[m0t]" if k == 0 else "s_add_u32 m0, m0, 0x1000"
        ops.append((m0, f"global_load_lds_dwordx4 %[vo{k}], %[tsrc]"))
    return ops


def slice_block(cur):
    """slice on set `cur`; builds set 1 - cur for the next slice (dword 1 - cur ... the next slice's parity)"""
    nxt = 1 - cur
    L = ["s_waitcnt lgkmcnt(%d)" % (2 if NOSYNC else 3)]          # fragment 0 is in; fragment 1 (2 reads) and the arrival add may be in flight
    dma = dma_ops()
    # SCHED 1: the build operations of all row fragments in order, one per VALU gap (cmp / cndmask stay adjacent in the
    # VALU stream: nothing else writes VCC)
    valu_slots = {}
    if SCHED and not NOBUILD:
        flat = [o for m in range(NM) for o in build_ops(nxt, m, nxt)]
        slots = [(F, g) for F in range(NF) for g in VALU_GAPS
                 if not (g == DMA_GAPS[1] and F in DMA_FRAGS) and not (F == CNT_CHECK_FRAG and g in (0, 1))
                 and not (F == CNT_READ_FRAG and g == CNT_GAP)]
        assert len(flat) <= len(slots), (len(flat), len(slots))
        for o, sl in zip(flat, slots):
            valu_slots.setdefault(sl, []).append(o)
    for F in range(NF):
        a = F % QG
        gaps = [[] for _ in range(NM)]
        if SCHED:
            for g in range(NM):
                gaps[g] += valu_slots.get((F, g), [])
        elif F < NM and not NOBUILD:
            ops = build_ops(nxt, F, nxt)
            for j, o in enumerate(ops):       # cmp / cndmask pairs in order, one operation per gap; the move rides along
                gaps[j if j < NM else NM - 2].append(o)
        if F == CNT_READ_FRAG and not NOSYNC:
            gaps[CNT_GAP if SCHED else RD_GAPS[0]].append(f"ds_read_b32 v{VCNT}, %[cnt]")
        if F == CNT_CHECK_FRAG and not NOSYNC:
            gaps[0] += [f"v_readfirstlane_b32 %[st], v{VCNT}", "s_cmp_ge_u32 %[st], %[tgt]"]
            gaps[1] += ["s_cbranch_scc1 .Lfwdw_go_%=", "s_mov_b32 %[sp], 0x400000", ".Lfwdw_poll_%=:",
                        f"ds_read_b32 v{VCNT}, %[cnt]", "s_waitcnt lgkmcnt(0)", f"v_readfirstlane_b32 %[st], v{VCNT}",
                        "s_cmp_ge_u32 %[st], %[tgt]", "s_cbranch_scc1 .Lfwdw_go_%=", "s_sub_u32 %[sp], %[sp], 1",
                        "s_cmp_lg_u32 %[sp], 0", "s_cbranch_scc1 .Lfwdw_poll_%=", "s_trap 2", ".Lfwdw_go_%=:"]
        # the reads of fragment F + 2 (the last two fragments read the first two of the next step's tile)
        rd = b_reads(F + 2, "%[lb]") if F + 2 < NF else b_reads(F + 2 - NF, "%[lbn]")
        if not NOLDS:
            gaps[RD_GAPS[0]].append(rd[0])
            gaps[RD_GAPS[1]].append(rd[1])
        if F in DMA_FRAGS and not NODMA:
            m0, ld = dma[DMA_FRAGS.index(F)]
            gaps[DMA_GAPS[0]].append(m0)      # (SALU: free beside the VALU operation of its gap)
            gaps[DMA_GAPS[1]].append(ld)
        if F + 1 < NF and not NOLDS:
            gaps[7].append("s_waitcnt lgkmcnt(2)")       # all but the two newest reads: fragment F + 1 is in
        for m in range(NM):
            L.append(f"v_smfmac_f32_16x16x64_f16 {acc(m, a)}, {vr(a_set(cur, m), 4)}, {vr(BR[RING[F]], 8)}, "
                     f"v{a_idx(cur, m)}")
            L.extend(gaps[m])
    L.append(f"s_waitcnt vmcnt({0 if NODMA else NVMEM})")
    if not NOSYNC:
        L += ["s_mov_b64 exec, 1", "ds_add_u32 %[cnt], %[one]", "s_mov_b64 exec, -1"]
    return L


def uprep_block():
    """G / IX of the next u from XN (landed: issued a whole u ago, and every block ends on a vmcnt), then the loads of the
    u after it.  %[arow] = byte offset of row fragment 0's row of this lane, %[rstride] = 16 rows, %[xsrc] = the
    alignment + 32 (u + 1)."""
    L = []
    for m in range(NM):
        for d in range(2):
            x, g, ix = XN + 2 * m + d, G + 2 * m + d, IX + 2 * m + d
            L += [f"v_add_u32_e32 v{T0}, 0x7f7f7f7f, v{x}",        # (state - 1) per byte: 0 -> 0xff, no carries (states < 128)
                  f"v_xor_b32_e32 v{T0}, 0x80808080, v{T0}",
                  f"v_lshrrev_b32_e32 v{g}, 2, v{T0}",
                  f"v_and_b32_e32 v{g}, 0x3f3f3f3f, v{g}",
                  f"v_and_b32_e32 v{T0}, 0x03030303, v{T0}",       # positions, one per byte ...
                  f"v_lshrrev_b32_e32 v{T1}, 4, v{T0}",
                  f"v_or_b32_e32 v{T1}, v{T1}, v{T0}",             # ... byte 0 = p0 | p1 << 4, byte 2 = p2 | p3 << 4
                  f"v_perm_b32 v{ix}, v{T1}, v{T1}, %[sel]"]       # index word: nibble p = position of pair p
    L.append(f"v_mov_b32_e32 v{T0}, %[arow]")
    for m in range(NM):
        L.append(f"global_load_dwordx2 {vr(XN + 2 * m, 2)}, v{T0}, %[xsrc]")
        if m + 1 < NM:
            L.append(f"v_add_u32_e32 v{T0}, %[rstride], v{T0}")
    return L


def first_loads_block():
    L = [f"v_mov_b32_e32 v{T0}, %[arow]"]
    for m in range(NM):
        L.append(f"global_load_dwordx2 {vr(XN + 2 * m, 2)}, v{T0}, %[xsrc]")
        if m + 1 < NM:
            L.append(f"v_add_u32_e32 v{T0}, %[rstride], v{T0}")
    L.append("s_waitcnt vmcnt(0)")
    return L


def prime_block():
    """set 0 for slice 0 of u = 0 (G / IX are set), the f16 constant, the first two B fragments of the first tile"""
    L = [f"v_mov_b32_e32 v{C1}, 0x3c00"]
    for m in range(NM):
        L += build_ops(0, m, 0)
    L += b_reads(0, "%[lbn]") + b_reads(1, "%[lbn]")
    return L


def issue_block():
    L = []
    for m0, ld in dma_ops():
        L += [m0, "s_nop 0", ld]
    return L


def zero_block():
    return [f"v_accvgpr_write_b32 a{i}, 0" for i in range(NACC)]


def cstr(lines):
    return " \\\n".join(f'    "{l}\\n\\t"' for l in lines)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.environ.get("FWDW_OUT") or os.path.join(here, "..", "evcouplings_amd", "csrc", "plm_fwd_asm.inc")
    clob = ", ".join([f'"v{i}"' for i in range(V_LO, 256)] + [f'"a{i}"' for i in range(NACC)])
    with open(out, "w") as fh:
        fh.write("// GENERATED by scripts/gen_fwd_asm.py -- do not edit; the register map is described there.\n")
        fh.write(f"#define PLM_FWDW_NM {NM}\n#define PLM_FWDW_QG {QG}\n#define PLM_FWDW_VLO {V_LO}\n")
        fh.write(f"#define PLM_FWDW_SLOT {SLOT}\n#define PLM_FWDW_NVMEM {NVMEM}\n")
        fh.write(f"#define PLM_FWDW_CLOBBERS {clob}, \"m0\", \"vcc\", \"scc\", \"memory\"\n")
        for name, lines in (("ZERO", zero_block()), ("ISSUE", issue_block()), ("LOADS", first_loads_block()),
                            ("UPREP", uprep_block()), ("PRIME", prime_block()), ("EVEN", slice_block(0)),
                            ("ODD", slice_block(1))):
            fh.write(f"#define PLM_FWDW_{name}_ASM \\\n" + cstr(lines) + "\n")
    print(f"wrote {os.path.normpath(out)}: slice block {len(slice_block(0))} instructions, {NF * NM} MFMAs")


if __name__ == "__main__":
    main()
