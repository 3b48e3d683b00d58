import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa
N, L, Q = 600, 24, 21
msa, _ = synthetic_msa(N, L, seed=31)
lj = plm.default_lambda_j(L, Q)
for mode in (None, "0", "1"):
    if mode is None: os.environ.pop("PLM_FWD_ACCURATE", None)
    else: os.environ["PLM_FWD_ACCURATE"] = mode
    res = plm.fit(msa, Q, lambda_h=0.01, lambda_j=lj, max_iter=3000, epsilon=2e-6, lbfgs_m=6)
    print("mode", mode, res["iters"], res["n_evals"], res["status_msg"], flush=True)
    print("   cond tail", ["%.2e" % r[2] for r in res["table"][-12:]], flush=True)
