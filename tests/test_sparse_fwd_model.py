"""The index arithmetic of the sparse-MFMA forward GEMM (k_expand tile layout + compressed one-hot fragments of k_fwd +
the operand pairing of v_smfmac_f32_16x16x64_f16 decoded on the GPU, profiles/r02_smfmac_probe.txt) restated in numpy:
together they must give HJ[s,i,a] = sum_{j != i} J_ij(a, x_sj).  Runs without a GPU; the GPU parity tests check the
kernels themselves against the oracle."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
from sparse_fwd_model import model  # noqa: E402


@pytest.mark.parametrize("L,Q", [(20, 5), (17, 4), (36, 21), (33, 20)])
def test_sparse_forward_index_arithmetic(L, Q):
    assert model(L, Q, N=6) < 1e-9


def test_probe_output_matches_the_pairing_the_model_assumes():
    """the committed probe output is the evidence for the pairing: pair p of A lane ga -> B lane gb = 2 (ga % 2) + p / 2,
    slots 8 (ga / 2) + 4 (p % 2) + position, position bits of compressed slot t at [2t+1 : 2t]"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_smfmac_probe.txt")
    import re
    seen = 0
    for line in open(path):
        m = re.match(r"A lane +(\d+) \(m +(\d+) ga (\d)\) slot (\d) idx 0x([0-9a-f]+) -> C row +(\d+), B \(gb (\d), slot +(\d+)\)", line)
        if not m:
            continue
        lane, mrow, ga, t, idx, crow, gb, slot = (int(m.group(k), 16 if k == 5 else 10) for k in range(1, 9))
        p = t // 2
        pos = (idx >> (2 * t)) & 3
        assert crow == mrow == lane % 16 and ga == lane // 16
        assert gb == 2 * (ga % 2) + p // 2
        assert slot == 8 * (ga // 2) + 4 * (p % 2) + pos
        seen += 1
    assert seen >= 160
