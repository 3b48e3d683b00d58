#!/usr/bin/env python3
"""Print the HIP-event kernel timings of the evaluation pipeline on the headline workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
ctx = plm.PlmContext(msa, q=21, max_iter=2, epsilon=1e-3)   # the production stop rule: three digit planes
ctx.set_weights(np.full(N, 0.9, np.float32)); ctx.marginals(pairs=False); ctx.set_x(None)
if os.environ.get("PLM_ZERO", "0") != "1":   # PLM_ZERO=1: time at the start point (J = 0: all-zero B operand)
    ctx.optimize()
km = ctx.time_kernels(reps=int(os.environ.get("PLM_REPS", 5)))
print(os.environ.get("PLM_HIP_LIB", "default").split("/")[-1], {k: round(v, 3) for k, v in km.items()})
