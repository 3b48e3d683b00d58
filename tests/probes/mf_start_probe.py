#!/usr/bin/env python3
"""GPU-side: does a mean-field start shorten the pseudo-likelihood fit?  The default fit starts at J = 0 and spends half of
its iterations reaching |g|/|x| < 1 (bench line: `first_time_cond_below`).  Mean-field DCA (row N4, one Cholesky of the
L(q-1) covariance matrix) gives couplings of the right structure in ~0.1 s.  This probe starts the headline fit from
c * J_mf (zero-sum gauge, a few scales c and pseudo-counts) and prints iterations / seconds next to the default start.
The optimum is the same (convex objective); only the path to it changes.
usage: mf_start_probe.py [CONFIG]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED

CONFIGS = {"c2": (20000, 200, 2), "c3": (100000, 300, 3), "headline": (50000, 300, 1), "c4": (50000, 500, 4),
           "c5": (30000, 600, 5)}
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
N, L, k = CONFIGS[name]
Q = 21
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + k)
iu, ju = np.triu_indices(L, 1)


def mf_couplings(pc):
    t = time.time()
    mf = plm.mean_field(msa, Q, pseudo_count=pc, want_fij=False, want_full=False, want_di=False)
    dt = time.time() - t
    J = mf["jij"].astype(np.float64)                  # [pairs, q, q], last row / column zero (reference-state gauge)
    J = J - J.mean(axis=1, keepdims=True) - J.mean(axis=2, keepdims=True) + J.mean(axis=(1, 2), keepdims=True)
    return J.astype(np.float32), dt


with plm.PlmContext(msa, q=Q, max_iter=1000, epsilon=1e-3) as ctx:
    ctx.reweight()
    ctx.marginals(pairs=False)
    ctx.set_x(None)
    x_def = ctx.get_x()
    t = time.time(); r = ctx.optimize(); dt = time.time() - t
    x_opt = ctx.get_x()
    print("%s default start: it=%d ev=%d %.3f s" % (name, r["iters"], r["n_evals"], dt), flush=True)
    Jopt = x_opt[L * Q:].reshape(-1, Q, Q)
    for pc in (0.5, 0.2, 0.05):
        Jmf, t_mf = mf_couplings(pc)
        # the scale that fits the optimum best (diagnostic) and the correlation with it
        c_best = float((Jmf * Jopt).sum() / (Jmf * Jmf).sum())
        corr = float((Jmf * Jopt).sum() / np.sqrt((Jmf * Jmf).sum() * (Jopt * Jopt).sum()))
        print("  pseudo-count %.2f: mean-field %.3f s (incl. transfers), best scale vs optimum %.3f, correlation %.4f, "
              "|J_mf| %.3f |J_opt| %.3f" % (pc, t_mf, c_best, corr, np.linalg.norm(Jmf), np.linalg.norm(Jopt)), flush=True)
        for c in (c_best, 0.5 * c_best, 1.0):
            x0 = x_def.copy()
            x0[L * Q:] = (c * Jmf).ravel()
            ctx.set_x(x0)
            fx0, _ = ctx.eval()
            ctx.set_x(x0)
            t = time.time(); r = ctx.optimize(); dt = time.time() - t
            print("    start %.3f * J_mf: f(start)=%.1f it=%d ev=%d %.3f s  cond=%.2e %s" % (
                c, fx0, r["iters"], r["n_evals"], dt, r["table"][-1][2], r["status_msg"][:30]), flush=True)
