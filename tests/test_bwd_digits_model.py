"""CPU model of the integer arithmetic of the int8 backward GEMM (kernels k_hpass / k_onehot_rt / k_bwd /
g_combine in evcouplings_amd/csrc/plm_kernels.hip): the digit decomposition of the fixed-point residuals, the
v_perm_b32 selectors that gather digit p of four values into a plane word, the 4 x 4 transpose over lane groups built
from v_permlane32_swap / v_permlane16_swap, the -128 one-hot operand and the recombination of the planes.  The GPU
tests check the kernels against the oracle; this pins the index / bit arithmetic they are written from, without a GPU
(the counterpart of tests/test_sparse_fwd_model.py for the forward GEMM)."""
import numpy as np
import pytest

QMAX3, QMAX4 = 8355000.0, 2138000000.0          # plm_internal.h PLM_R_QMAX3 / PLM_R_QMAX4


def digits_of(v, four):
    """kernel digits_of(): R = rint(v) (ties to even), (R + bias) ^ bias with bias 0x808080 / 0x80808080 in u32 arithmetic"""
    bias = np.uint32(0x80808080 if four else 0x00808080)
    R = np.rint(np.asarray(v, dtype=np.float32)).astype(np.int64)
    return ((R.astype(np.uint32) + bias) ^ bias), R


def v_perm_b32(s0, s1, sel):
    """V_PERM_B32: result byte i = byte sel.byte[i] of the 64-bit value {s0 (bytes 4-7), s1 (bytes 0-3)}; 0x0c = 0x00"""
    src = [(s1 >> (8 * k)) & 0xff for k in range(4)] + [(s0 >> (8 * k)) & 0xff for k in range(4)]
    out = np.zeros_like(s0)
    for i in range(4):
        k = (sel >> (8 * i)) & 0xff
        out |= (np.zeros_like(s0) if k == 0x0c else src[k]) << (8 * i)
    return out


def planes_of4(x0, x1, x2, x3):
    t01, t23 = v_perm_b32(x1, x0, 0x05010400), v_perm_b32(x3, x2, 0x05010400)
    u01, u23 = v_perm_b32(x1, x0, 0x07030602), v_perm_b32(x3, x2, 0x07030602)
    return [v_perm_b32(t23, t01, 0x05040100), v_perm_b32(t23, t01, 0x07060302),
            v_perm_b32(u23, u01, 0x05040100), v_perm_b32(u23, u01, 0x07060302)]


def as_i8(b):
    return np.asarray(b, dtype=np.uint8).view(np.int8).astype(np.int64)


@pytest.mark.parametrize("four", [False, True])
def test_digits_reconstruct_the_fixed_point_residual(four):
    rng = np.random.default_rng(1)
    q = QMAX4 if four else QMAX3
    v = np.concatenate([rng.uniform(-q, q, 200000), [q, -q, 0.0, 0.5, -0.5, 1.5, 2.5, -127.5, 128.5, 32768.5]]).astype(np.float32)
    x, R = digits_of(v, four)
    d = [as_i8((x >> np.uint32(8 * k)) & np.uint32(0xff)) for k in range(4)]
    n = 4 if four else 3
    assert all((-128 <= dk).all() and (dk <= 127).all() for dk in d[:n])
    if not four:
        assert ((x >> np.uint32(24)) == 0).all()                          # nothing spills into a fourth byte
    np.testing.assert_array_equal(sum(d[k] * 256 ** k for k in range(n)), R)
    # ties go to even: unbiased
    assert digits_of(np.float32(0.5), four)[1] == 0 and digits_of(np.float32(1.5), four)[1] == 2


def test_plane_words_gather_one_digit_of_four_values():
    rng = np.random.default_rng(2)
    x = rng.integers(0, 2 ** 32, size=(4, 1000), dtype=np.uint64).astype(np.uint32)
    pl = planes_of4(*x)
    for p in range(4):
        for k in range(4):
            np.testing.assert_array_equal((pl[p] >> np.uint32(8 * k)) & np.uint32(0xff), (x[k] >> np.uint32(8 * p)) & np.uint32(0xff))


def test_lane_group_transpose_gives_every_row_one_state():
    """X[k] (k = state of the group of four) as rows over the lane groups g = 0..3; after permlane32_swap on (X0, X2),
    (X1, X3) and permlane16_swap on the results, row g holds state g's dwords of source rows 0..3 in order."""
    def swap32(a, b):      # v_permlane32_swap: upper half (rows 2, 3) of a <-> lower half (rows 0, 1) of b
        return [a[0], a[1], b[0], b[1]], [a[2], a[3], b[2], b[3]]

    def swap16(a, b):      # v_permlane16_swap: odd rows of a <-> even rows of b
        return [a[0], b[0], a[2], b[2]], [a[1], b[1], a[3], b[3]]
    X = [[(k, g) for g in range(4)] for k in range(4)]                     # element = (state k, source row g)
    s02a, s02b = swap32(X[0], X[2])
    s13a, s13b = swap32(X[1], X[3])
    t0a, t0b = swap16(s02a, s13a)
    t1a, t1b = swap16(s02b, s13b)
    for g in range(4):
        assert [t0a[g], t0b[g], t1a[g], t1b[g]] == [(g, 0), (g, 1), (g, 2), (g, 3)]


@pytest.mark.parametrize("four", [False, True])
def test_planes_times_minus128_onehot_recombine_to_the_weighted_sum(four):
    """G_p = sum_s (-128 [x_s = b]) d_p(s) in integers; g = gscale (G_0 + 256 G_1 + ...) with gscale = wmax / (QMAX * -128)
    must be sum_s [x_s = b] r_s up to the quantisation (half a unit of 1 / rscale per term)."""
    rng = np.random.default_rng(3)
    n, wmax = 5000, 0.7
    r = (rng.uniform(-1, 1, n) * wmax * rng.random(n)).astype(np.float32)
    states = rng.integers(0, 21, n)
    q = QMAX4 if four else QMAX3
    rscale = np.float32(q / wmax)
    x, R = digits_of(r * rscale, four)
    planes = 4 if four else 3
    d = [as_i8((x >> np.uint32(8 * k)) & np.uint32(0xff)) for k in range(planes)]
    gscale = wmax / (q * -128.0)
    for b in (0, 7, 20):
        onehot = np.where(states == b, -128, 0)
        # the bound the kernel relies on: |sum| <= 128 * 127 * K < 2^31
        G = [int((onehot * dk).sum()) for dk in d]
        assert all(abs(g) < 2 ** 31 for g in G)
        got = gscale * sum(G[k] * 256.0 ** k for k in range(planes))
        want = float(r[states == b].astype(np.float64).sum())
        assert abs(got - want) <= 0.5 * (states == b).sum() / float(rscale) + 1e-6 * abs(want)


def test_onehot16_marks_matches_with_minus128():
    """k_bwd's expansion: y = (x ^ b) + 0x7f7f7f7f sets bit 7 of every byte that differs from b (states < 128: no
    carries), ~y & 0x80808080 keeps 0x80 = -128 (int8) exactly at the matching bytes."""
    rng = np.random.default_rng(4)
    xb = rng.integers(0, 128, size=(1000, 4)).astype(np.uint32)
    x = xb[:, 0] | (xb[:, 1] << 8) | (xb[:, 2] << 16) | (xb[:, 3] << 24)
    for b in (0, 1, 20, 127):
        bb = np.uint32(b * 0x01010101)
        y = ((x ^ bb).astype(np.uint64) + 0x7f7f7f7f).astype(np.uint32)
        a = ~y & np.uint32(0x80808080)
        for k in range(4):
            np.testing.assert_array_equal(as_i8((a >> np.uint32(8 * k)) & np.uint32(0xff)), np.where(xb[:, k] == b, -128, 0))


def test_residual_fragment_addressing_writer_and_reader_agree():
    """The Rt layout [plane][128-sequence step][column fragment][half][64 lanes][16 B] as k_hpass writes it (a workgroup =
    256 sequences x 16 sites, wave w = 32 sequences, two halves m of 16, lane (g = lane / 16, r = lane % 16) holding the
    4 sequences 4 g .. 4 g + 3 of the half) against how k_bwd reads it (K step ss, half H, fragment lane (G, c) = the 16
    consecutive sequences 128 ss + 64 H + 16 G + e of site c).  Every (plane, sequence, column fragment, site) byte
    must have exactly one writer, at the address the reader expects."""
    Np, nnfl, Q, nplanes = 512, 2 * 5, 5, 3
    nst128 = Np // 128
    expect = {}
    # reader: byte address -> (plane, sequence, nf, site)
    for p in range(nplanes):
        for ss in range(nst128):
            for nf in range(nnfl):
                for H in range(2):
                    for lane in range(64):
                        G, c = lane >> 4, lane & 15
                        for e in range(16):
                            addr = ((((p * nst128 + ss) * nnfl + nf) * 2 + H) * 1024) + lane * 16 + e
                            expect[addr] = (p, 128 * ss + 64 * H + 16 * G + e, nf, c)
    # writer (k_hpass): tile stile of 256 sequences, wave, half m, lane (g, r), state a of site block b16l
    seen = {}
    for stile in range(Np // 256):
        for b16l in range(nnfl // Q):
            for wave in range(8):
                s64 = stile * 4 + (wave >> 1)
                for m in range(2):
                    Gq = 2 * (wave & 1) + m
                    for lane in range(64):
                        g, r = lane >> 4, lane & 15
                        for a in range(Q):
                            nf = b16l * Q + a
                            for p in range(nplanes):
                                base = ((((p * nst128 + (s64 >> 1)) * nnfl + nf) * 2 + (s64 & 1)) * 1024) + (Gq * 16 + r) * 16
                                for reg in range(4):
                                    s = stile * 256 + wave * 32 + 16 * m + 4 * g + reg
                                    addr = base + 4 * g + reg
                                    assert addr not in seen
                                    seen[addr] = (p, s, nf, r)
    assert seen == expect
