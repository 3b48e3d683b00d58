#!/usr/bin/env python3
"""GPU-side: the smallest field-subproblem gradient the solver reaches (PLM_DEBUG_VP lines of a fit with an unreachable
tolerance), to place the floor of its tolerance.  usage: PLM_DEBUG_VP=1 PLM_VP_FLOOR=0 vp_floor_probe.py 2> log"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
for (N, L) in ((500, 56), (4000, 100), (20000, 200), (50000, 300)):
    msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
    sys.stderr.write("[probe] N=%d L=%d\n" % (N, L)); sys.stderr.flush()
    res = plm.fit(msa, 21, max_iter=25, epsilon=1e-12, want_fij=False)
    sys.stderr.write("[probe] n_eff=%.1f iters=%d evals=%d opt=%.3fs\n" % (res["n_eff"], res["iters"], res["n_evals"], res["seconds"]["optimize"]))
