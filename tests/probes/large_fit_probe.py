"""Whole fits far above the BASELINE sizes on one MI355X (capability data points, no oracle): a very deep alignment, a long
one, a mid-size one.  usage: large_fit_probe.py [N L ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa

args = [int(v) for v in sys.argv[1:]]
shapes = list(zip(args[0::2], args[1::2])) or [(500000, 300), (50000, 1000), (200000, 100)]
for N, L in shapes:
    t = time.time()
    msa, planted = synthetic_msa(N, L, seed=N + L)
    t_gen = time.time() - t
    t = time.time()
    r = plm.fit(msa, q=21, max_iter=1500, epsilon=1e-3, want_fij=False)
    dt = time.time() - t
    cn = r["cn"]
    iu = np.triu_indices(L, 6)
    order = np.argsort(-cn[iu])[: len(planted)]
    top = {(int(iu[0][k]), int(iu[1][k])) for k in order}
    hit = sum((min(i, j), max(i, j)) in top for i, j in planted)
    print("N=%d L=%d: n_eff %.0f, %d iterations / %d evaluations, %s; plm.fit %.2f s (reweight %.3f, optimise %.2f), "
          "%d of %d planted pairs among the top %d CN scores (alignment generated in %.0f s)"
          % (N, L, r["n_eff"], r["iters"], r["n_evals"], r["status_msg"][:50], dt, r["seconds"]["reweight"],
             r["seconds"]["optimize"], hit, len(planted), len(planted), t_gen), flush=True)
