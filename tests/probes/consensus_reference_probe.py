#!/usr/bin/env python3
"""GPU-side probe (uses the f64 oracle): would a per-site CONSENSUS reference state shrink the error of the forward GEMM?
k_fwd accumulates differences to state 0 (the gap); the running sum carries -C_i(a) = -sum_j J_ij(a, 0), several times
the potential it ends up as, and its f32 roundings are the evaluation's error at scale.  Relabelling
the states per site (most frequent state -> label 0) gives a mathematically equivalent problem whose state 0 IS the
consensus: fit both, compare |g_hip - g_f64| / |x| at the stop point.  No library change needed for the measurement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
from oracle.oracle import Oracle
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300)); Q = 21
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + int(os.environ.get("PLM_SEED", 1)))
orc = Oracle("f64"); orc.set_num_threads(int(os.environ.get("PLM_THREADS", 16)))
lj = plm.default_lambda_j(L, Q)
# per-site relabelling: states in order of decreasing count
relabelled = np.empty_like(msa)
for j in range(L):
    order = np.argsort(-np.bincount(msa[:, j], minlength=Q), kind="stable")
    lut = np.empty(Q, np.int8); lut[order] = np.arange(Q, dtype=np.int8)
    relabelled[:, j] = lut[msa[:, j]]
for name, m in (("gap reference (as shipped)", msa), ("consensus reference (relabelled)", relabelled)):
    with plm.PlmContext(m, Q, max_iter=int(os.environ.get("PLM_MAXIT", 600)), epsilon=1e-3) as ctx:
        w, _, _ = ctx.reweight(); ctx.marginals(pairs=False); ctx.set_x(None)
        r = ctx.optimize(); x = ctx.get_x()
    fx, nll, g = plm.evaluate(m, w, Q, 0.01, lj, x)
    fxo, nllo, go = orc.eval(m, w.astype(np.float64), Q, 0.01, lj, x.astype(np.float64))
    e = g - go
    print("%-34s %3d iterations / %3d evaluations (%s)  |x| %.1f  err/|x| %.3e  oracle cond %.3e  fx %.4f" % (
        name, r["iters"], r["n_evals"], "converged" if r["status"] == 0 else "status %d" % r["status"],
        np.linalg.norm(x), np.linalg.norm(e) / np.linalg.norm(x), np.linalg.norm(go) / np.linalg.norm(x), fx))
