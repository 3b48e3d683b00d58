#!/usr/bin/env python3
"""GPU-side: fit time / iterations as a function of the field-solver tolerance floor (PLM_VP_FLOOR)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
for (N, L, seed) in ((50000, 300, 1), (100000, 300, 3), (20000, 200, 2)):
    msa, _ = synthetic_msa(N, L, seed=BASE_SEED + seed)
    for fl in sys.argv[1:]:   # coefficient of sqrt(N_eff L q)
        os.environ["PLM_VP_FLOOR"] = fl
        res = plm.fit(msa, 21, max_iter=1500, epsilon=1e-3, want_fij=False)
        print("N=%d L=%d floor=%s: iters=%d evals=%d status=%d cond=%.3e opt=%.2fs" % (
            N, L, fl, res["iters"], res["n_evals"], res["status"], res["table"][-1][2], res["seconds"]["optimize"]), flush=True)
