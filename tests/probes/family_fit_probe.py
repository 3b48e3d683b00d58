#!/usr/bin/env python3
"""GPU-side: the default fit on protein-family-like alignments (evcouplings_amd.synthetic.family_msa: clades, N_eff << N,
conserved columns, indel runs, duplicates) -- does it converge, in how many iterations, and does the f64 oracle agree that
the shipped point meets the stop rule?  usage: family_fit_probe.py [N L depth a b]..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import family_msa
from oracle.oracle import Oracle

cases = [(20000, 150, 5, 1.0, 15.0), (20000, 150, 4, 1.0, 25.0), (50000, 300, 5, 1.0, 15.0)]
orc = Oracle("f64")
for (N, L, depth, a, b) in cases:
    msa, _ = family_msa(N, L, seed=11, depth=depth, row_mut=(a, b))
    for gaps in (False, True):
        qm = 20 if gaps else 21
        with plm.PlmContext(msa, q=21, max_iter=2000, epsilon=1e-3, ignore_gaps=gaps) as ctx:
            t = time.time(); w, _, n_eff = ctx.reweight(); t_rw = time.time() - t
            ctx.marginals(pairs=False)
            ctx.set_x(None)
            t = time.time(); r = ctx.optimize(); dt = time.time() - t
            st = ctx.solver_stats()
            x = ctx.get_x()
            lam = ctx.lambda_j
        line = "N=%d L=%d depth=%d mut=(%g,%g) gaps=%d: n_eff=%.0f reweight %.1f ms | it=%d ev=%d %.2f s passes/ev=%.2f cont=%d | %s" % (
            N, L, depth, a, b, gaps, n_eff, 1e3 * t_rw, r["iters"], r["n_evals"], dt, st["passes_per_evaluation"],
            st["chains_continued_by_host"], r["status_msg"][:60])
        if N * L * L <= 20000 * 150 * 150:
            fn = orc.eval_gaps if gaps else orc.eval
            fx, nll, g = fn(msa, w.astype(np.float64), 21, 0.01, lam, x.astype(np.float64))
            line += " | oracle cond %.3e" % (np.linalg.norm(g) / max(1.0, np.linalg.norm(x)))
        print(line, flush=True)
