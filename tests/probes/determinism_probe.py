#!/usr/bin/env python3
"""GPU-side probe: is one objective+gradient evaluation bit-reproducible?  (many reps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for (N, L) in ((50000, 300), (20000, 200)):
    msa, _ = synthetic_msa(N, L, seed=7)
    ctx = plm.PlmContext(msa, q=21, max_iter=15, epsilon=1e-3)
    ctx.reweight(); ctx.marginals(pairs=False); ctx.set_x(None)
    ctx.optimize()
    x = ctx.get_x()
    ref, bad = None, 0
    for rep in range(reps):
        ctx.set_x(x)
        fx, nll = ctx.eval()
        g = ctx.get_g()
        if ref is None:
            ref = (fx, nll, g)
        elif fx != ref[0] or not np.array_equal(g, ref[2]):
            bad += 1
            dg = g - ref[2]
            print("  N=%d L=%d rep%d DIFFERS: dfx=%.3e max|dg|=%.3e ndiff=%d" % (N, L, rep, fx - ref[0], np.abs(dg).max(), (dg != 0).sum()))
    print("N=%d L=%d: %d of %d repeats differ from the first" % (N, L, bad, reps - 1))
    ctx.close()
