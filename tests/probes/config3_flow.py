#!/usr/bin/env python3
"""GPU-side: the flow of tests/test_gpu_scale.py's fixture on one configuration (20 iterations, then a resumed fit to the
stop rule) with the line-search diagnostics of PLM_DEBUG=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 100000)); L = int(os.environ.get("PLM_L", 300)); k = int(os.environ.get("PLM_SEED", 3))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + k)
with plm.PlmContext(msa, 21, max_iter=20, epsilon=1e-3) as ctx:
    ctx.reweight(); ctx.marginals(pairs=False); ctx.set_x(None)
    r = ctx.optimize(); print("first:", r["iters"], r["n_evals"], r["status_msg"])
    ctx.set_options(max_iter=int(os.environ.get("PLM_MAXIT", 3000)), epsilon=1e-3)
    t = time.time(); r = ctx.optimize()
    print("resumed: iters=%d evals=%d %s %.2fs" % (r["iters"], r["n_evals"], r["status_msg"], time.time() - t))
    for row in r["table"][-5:]:
        print("  it=%d cond=%.3e fx=%.4f" % (row[0], row[2], row[3]))
