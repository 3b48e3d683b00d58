#!/usr/bin/env python3
"""Workload for rocprofv3 counter passes: a short fit on the headline MSA (the default variable-projection
pipeline: k_fwd store mode, k_hpass, k_hsolve, k_bwd, k_assemble) followed by a few joint evaluations (k_fwd solver
mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
ctx = plm.PlmContext(msa, q=21, max_iter=int(os.environ.get("PLM_ITERS", 12)), epsilon=1e-3)
if os.environ.get("PLM_REWEIGHT", "0") == "1":
    ctx.reweight()
else:
    ctx.set_weights(np.full(N, 0.9, np.float32))
ctx.marginals(pairs=False); ctx.set_x(None); ctx.optimize()
for _ in range(int(os.environ.get("PLM_EVALS", 2))):
    ctx.eval()
print("done")
