"""k_bwd_w and k_fwd_w run their K loops as generated assembly blocks (scripts/gen_bwd_asm.py, scripts/gen_fwd_asm.py ->
evcouplings_amd/csrc/plm_*_asm.inc).  Without a GPU: the committed .inc files are what the generators write, and in the
ISA hipcc produces for the two kernels nothing outside the blocks touches the registers the blocks keep live across the
C++ code between them (scripts/check_bwd_asm.py).  The GPU parity tests compare the kernels' results bit for bit with the
compiler-allocated k_bwd / k_fwd."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "evcouplings_amd", "csrc")


@pytest.mark.parametrize("gen,var,inc", [("gen_bwd_asm.py", "BWDW_OUT", "plm_bwd_asm.inc"),
                                         ("gen_fwd_asm.py", "FWDW_OUT", "plm_fwd_asm.inc")])
def test_committed_blocks_are_what_the_generators_write(tmp_path, gen, var, inc):
    out = str(tmp_path / inc)
    env = {k: v for k, v in os.environ.items() if not k.startswith(("BWDW_", "FWDW_"))}   # no experiment switches
    env[var] = out
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", gen)], check=True, env=env, capture_output=True)
    assert open(out).read() == open(os.path.join(CSRC, inc)).read()


def test_ring_of_b_fragments_never_overwrites_a_fragment_in_use():
    """k_fwd_w reads B fragments two ahead into a ring of four register octets: any three consecutive fragments -- across
    the block boundary too -- must sit in different entries"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import gen_fwd_asm as g
    finally:
        sys.path.pop(0)
    n = len(g.RING)
    assert n == g.NF
    for i in range(n):
        assert len({g.RING[i], g.RING[(i + 1) % n], g.RING[(i + 2) % n]}) == 3
    # register map: tuples 64-bit aligned, nothing overlaps, everything inside the clobbered range
    used = []
    for s in range(2):
        for m in range(g.NM):
            assert g.a_set(s, m) % 2 == 0
            used += list(range(g.a_set(s, m), g.a_set(s, m) + 4)) + [g.a_idx(s, m)]
    for base in (g.G, g.IX, g.XN):
        used += list(range(base, base + 2 * g.NM))
    for b in g.BR:
        assert b % 2 == 0
        used += list(range(b, b + 8))
    used += [g.C1, g.VCNT, g.T0, g.T1]
    assert len(used) == len(set(used)) and min(used) >= g.V_LO and max(used) <= 255
    assert g.NACC <= 256


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_compiler_code_stays_out_of_the_blocks_registers(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    asm = str(tmp_path / "kernels.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", asm,
                    os.path.join(CSRC, "plm_kernels.hip")], check=True, capture_output=True, timeout=600)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_bwd_asm.py"), asm], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "k_bwd_w<21>" in run.stdout and "k_fwd_w" in run.stdout


def test_fragment_build_bit_tricks_equal_the_plain_formula():
    """k_fwd builds the compressed one-hot fragment of a slice from four alignment bytes with compares and shifts per pair
    (plm_kernels.hip: value 1.0 where (x - 1) >> 2 equals the slice's state group, index nibble (x - 1) & 3); k_fwd_w's
    assembly does the same with word-wide tricks: a byte-wise decrement without carries, (x + 0x7f7f7f7f) ^ 0x80808080,
    the groups as (y >> 2) & 0x3f3f3f3f compared byte by byte, the positions packed from bytes to nibbles with
    t = p | p >> 4 and a byte permute.  Restated here in integer arithmetic for every state byte 0..127."""
    import numpy as np
    rng = np.random.default_rng(0)
    m32 = 0xFFFFFFFF
    cases = [np.arange(4 * k, 4 * k + 4) for k in range(32)] + [rng.integers(0, 128, 4) for _ in range(2000)]
    for xs in cases:
        xs = [int(v) for v in xs]
        word = sum(x << (8 * p) for p, x in enumerate(xs))
        y = ((word + 0x7F7F7F7F) & m32) ^ 0x80808080
        for p, x in enumerate(xs):                      # byte-wise x - 1, with 0 -> 0xff
            assert (y >> (8 * p)) & 0xFF == (x - 1) & 0xFF
        grp = (y >> 2) & 0x3F3F3F3F
        pos = y & 0x03030303
        t = (pos | (pos >> 4)) & m32
        idx = (t & 0xFF) | (((t >> 16) & 0xFF) << 8)    # v_perm_b32 selector 0x0c0c0200: byte 0, byte 2, zero, zero
        for p, x in enumerate(xs):
            u = (x - 1) & m32
            assert (idx >> (4 * p)) & 0xF == u & 3      # the kernel's `((x - 1) & 3) << (4 * pp)`; upper index bits 0
            for kg in range(5):
                assert (((grp >> (8 * p)) & 0xFF) == kg) == ((u >> 2) == kg)
