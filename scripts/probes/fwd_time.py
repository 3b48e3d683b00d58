#!/usr/bin/env python3
"""GPU-side: the forward / backward GEMM launched alone at the headline (plm_ctx_time_kernels), for kernel A/B builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
msa, _ = synthetic_msa(50000, 300, seed=BASE_SEED + 1)
x0 = (0.01 * np.random.default_rng(1).normal(size=plm.n_params(300, 21))).astype(np.float32)
with plm.PlmContext(msa, q=21, max_iter=20, epsilon=1e-3) as c:
    c.set_weights(np.full(50000, 0.9, np.float32)); c.set_x(x0)
    km = c.time_kernels(reps=10)
print(os.path.basename(os.environ.get("PLM_HIP_LIB", "default")), {k: round(km[k], 3) for k in ("forward", "backward", "expand", "assemble")})
