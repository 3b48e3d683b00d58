#!/usr/bin/env python3
"""GPU-side: random rounding noise of one evaluation, measured without an oracle: the same point evaluated with two
power-of-two pre-scales of the coupling operand (PLM_JEXP_BIAS 0 / -1 move every hi/lo split point and every f32
rounding of the forward GEMM, the mathematics is unchanged).  Prints |g0 - g1| / |x| and |fx0 - fx1| / fx at the
point a fit reached after PLM_ITERS iterations (joint evaluation, plm_ctx_eval).  A/B between libraries with
PLM_HIP_LIB (e.g. a -DPLM_SPARSE_FWD=0 build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
with plm.PlmContext(msa, q=21, max_iter=int(os.environ.get("PLM_ITERS", 120)), epsilon=1e-12) as ctx:
    ctx.reweight(); ctx.marginals(pairs=False); ctx.set_x(None)
    ctx.optimize()
    x = ctx.get_x()
    out = []
    for bias in ("0", "-1", "0"):
        os.environ["PLM_JEXP_BIAS"] = bias
        ctx.set_x(x)
        fx, nll = ctx.eval()
        out.append((fx, ctx.get_g()))
    xn = np.linalg.norm(x)
    print(os.environ.get("PLM_HIP_LIB", "default").split("/")[-1],
          "|x| %.2f  |g| / |x| %.3e   bias 0 vs -1: |dg| / |x| %.3e  |dfx| / fx %.3e   bias 0 vs 0: |dg| / |x| %.3e" % (
              xn, np.linalg.norm(out[0][1]) / xn, np.linalg.norm(out[0][1] - out[1][1]) / xn,
              abs(out[0][0] - out[1][0]) / abs(out[0][0]), np.linalg.norm(out[0][1] - out[2][1]) / xn))
