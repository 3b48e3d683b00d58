#!/usr/bin/env python3
"""GPU-side: the two kinds of chain position timed alone at the headline (plm_ctx_time_field_positions), for kernel A/B
runs (PLM_HIP_LIB) and under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
rng = np.random.default_rng(1)
x0 = (0.01 * rng.normal(size=plm.n_params(L, 21))).astype(np.float32)
with plm.PlmContext(msa, q=21, max_iter=20, epsilon=1e-3) as c:
    c.set_weights(np.full(N, 0.9, np.float32))
    c.set_x(x0)
    c.time_kernels(reps=1)
    print(os.environ.get("PLM_HIP_LIB", "default"), c.time_field_positions(reps=10))
