#!/usr/bin/env python3
"""GPU-side: wall time of everything around the optimiser in one run_plmc-style fit (headline shape): context creation,
reweighting, marginals with pair frequencies, parameter download, scores, teardown -- and plm.fit as a whole."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N, L = int(os.environ.get("PLM_N", 50000)), int(os.environ.get("PLM_L", 300))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
plm.reweight(msa[:1000], 0.8)     # library + device warm
for gaps in (False, True):
    T = {}
    t = time.time(); ctx = plm.PlmContext(msa, q=21, max_iter=20, epsilon=1e-3, ignore_gaps=gaps); T["create"] = time.time() - t
    t = time.time(); ctx.reweight(); T["reweight"] = time.time() - t
    t = time.time(); ctx.marginals(pairs=True); T["marginals+fij"] = time.time() - t
    t = time.time(); ctx.set_x(None); T["set_x"] = time.time() - t
    t = time.time(); ctx.optimize(); T["optimize(20)"] = time.time() - t
    t = time.time(); ctx.get_x(); T["get_x"] = time.time() - t
    t = time.time(); ctx.scores(); T["scores"] = time.time() - t
    t = time.time(); ctx.close(); T["close"] = time.time() - t
    print("gaps=%d  " % gaps + "  ".join("%s %.1f ms" % (k, 1e3 * v) for k, v in T.items()), flush=True)
    t = time.time(); r = plm.fit(msa, 21, max_iter=20, epsilon=1e-3, ignore_gaps=gaps); dt = time.time() - t
    print("gaps=%d  plm.fit(20 iterations) %.1f ms wall; library: %s" % (gaps, 1e3 * dt, {k: round(1e3 * v, 1) for k, v in r["seconds"].items()}), flush=True)
