#!/usr/bin/env python3
"""GPU-side: headline fit trajectory (cond every 100 iterations) and line-search diagnostics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300)); m = int(os.environ.get("PLM_M", 6))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
t = time.time()
res = plm.fit(msa, 21, max_iter=int(os.environ.get("PLM_MAXIT", 1500)), epsilon=float(os.environ.get("PLM_EPS", 1e-3)), lbfgs_m=m, want_fij=False)
print("m=%d: iters=%d evals=%d status=%d (%s) %.2fs" % (m, res["iters"], res["n_evals"], res["status"], res["status_msg"], time.time() - t))
tab = res["table"]
cn_prev = None
for r in tab:
    if r[0] in (1, 10, 50, 100) or r[0] % 250 == 0 or r[0] == res["iters"]:
        print("  it=%5d t=%7.3f cond=%.3e fx=%.4f |h|=%.3f |e|=%.3f" % (r[0], r[1], r[2], r[3], r[5], r[6]))
