#!/usr/bin/env python3
"""CPU model of the sparse-MFMA forward GEMM's index arithmetic (plm_kernels.hip: k_expand / k_fwd with PLM_SPARSE_FWD):
the tile layout k_expand writes, the compressed one-hot fragments k_fwd builds, and the operand pairing of
v_smfmac_f32_16x16x64_f16 as decoded on the GPU (profiles/r02_smfmac_probe.txt) -- together they must reproduce
HJ[s,i,a] = sum_{j != i} J_ij(a, x_sj).  No GPU needed: run it after touching either kernel's indexing."""
import numpy as np


def model(L, Q, N=16, seed=0):
    rng = np.random.default_rng(seed)
    NG = (Q + 2) // 4                                      # groups of 4 over the states 1 .. Q-1
    nu = (L + 31) // 32
    x = rng.integers(0, Q, size=(N, L))
    J = rng.normal(size=(L, Q, L, Q))                      # J[i,a,j,b], any values: the model checks indexing only
    want = np.zeros((N, L, Q))
    for s in range(N):
        for j in range(L):
            want[s] += J[:, :, j, x[s, j]]
        for i in range(L):
            want[s, i] -= J[i, :, i, x[s, i]]              # j != i
    xpad = np.full((N, nu * 32), 127)
    xpad[:, :L] = x
    got = np.zeros((N, L, Q))
    for I in range((L + 15) // 16):                        # column block: sites i = 16 I + n
        for u in range(nu):
            for ci in range(2 * NG):
                # ---- k_expand: tile[a][lane = (n, gb)][slot 0..15]  (plane split omitted: values stay f64)
                tile = np.zeros((Q, 64, 16))
                for a in range(Q):
                    for lane in range(64):
                        gb, n = lane >> 4, lane & 15
                        i = 16 * I + n
                        for half in range(2):
                            for e8 in range(8):
                                ga = 2 * half + (gb >> 1)
                                pp = 2 * (gb & 1) + (e8 >> 2)
                                e = e8 & 3
                                gl = 4 * ci + pp
                                s8, sg = gl % 8, gl // 8
                                b, j = 4 * sg + e + 1, 32 * u + 8 * ga + s8
                                if b < Q and i < L and j < L and i != j:
                                    tile[a, lane, 8 * half + e8] = J[i, a, j, b] - J[i, a, j, 0]
                # ---- k_fwd: compressed A of row s in lane (m, ga); hardware pairing of pair p -> B (gb, slots)
                for s in range(N):
                    for ga in range(4):
                        for pp in range(4):
                            gl = 4 * ci + pp
                            s8, sg = gl % 8, gl // 8
                            xs = xpad[s, 32 * u + 8 * ga + s8]
                            if ((xs - 1) & 0xffffffff) >> 2 != sg:
                                continue                                   # compressed value 0 (incl. x = 0 and padding)
                            pos = (xs - 1) & 3
                            gb = 2 * (ga % 2) + pp // 2                     # decoded operand pairing
                            slot = 8 * (ga // 2) + 4 * (pp % 2) + pos
                            for n in range(16):
                                i = 16 * I + n
                                if i < L:
                                    got[s, i, :] += tile[:, 16 * gb + n, slot]
    for i in range(L):                                     # k_fwd_ref: the reference-state constant
        got[:, i, :] += sum(J[i, :, j, 0] for j in range(L) if j != i)[None, :]
    err = np.abs(got - want).max()
    return err


if __name__ == "__main__":
    for (L, Q) in ((40, 21), (33, 20), (20, 5), (17, 4), (70, 21)):
        e = model(L, Q)
        print("L=%d Q=%d  max |model - direct| = %.3e" % (L, Q, e))
        assert e < 1e-9
    print("ok")
