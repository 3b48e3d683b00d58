#!/usr/bin/env python3
"""GPU-side probe (uses the f64 oracle): how much of |g_hip - g_f64| at the stop point of a headline fit is the rounding
of the couplings into the forward GEMM's f16 hi + lo operand planes (22 significant bits)?  Evaluate both sides at the
fit's x and at x snapped to a grid (multiples of 2^-12) on which every coupling DIFFERENCE is exactly representable in
hi + lo: if the operand rounding is the culprit the error collapses at the snapped point."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
from oracle.oracle import Oracle
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300)); Q = 21
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
orc = Oracle("f64"); orc.set_num_threads(int(os.environ.get("PLM_THREADS", 16)))
lj = plm.default_lambda_j(L, Q)
with plm.PlmContext(msa, Q, max_iter=int(os.environ.get("PLM_MAXIT", 400)), epsilon=1e-3) as ctx:
    w, _, _ = ctx.reweight(); ctx.marginals(pairs=False); ctx.set_x(None)
    r = ctx.optimize(); x = ctx.get_x()
    print("fit:", r["iters"], r["n_evals"], r["status_msg"])
for name, xx in (("fit point", x), ("snapped to 2^-12", (np.round(x.astype(np.float64) * 4096) / 4096).astype(np.float32)),
                 ("snapped to 2^-16", (np.round(x.astype(np.float64) * 65536) / 65536).astype(np.float32))):
    fx, nll, g = plm.evaluate(msa, w, Q, 0.01, lj, xx)
    fxo, nllo, go = orc.eval(msa, w.astype(np.float64), Q, 0.01, lj, xx.astype(np.float64))
    e = g - go
    nh = L * Q
    print("%-18s |x| %.2f  |g64|/|x| %.3e  err/|x|: total %.3e  fields %.3e  couplings %.3e   fx rel %.2e" % (
        name, np.linalg.norm(xx), np.linalg.norm(go) / np.linalg.norm(xx), np.linalg.norm(e) / np.linalg.norm(xx),
        np.linalg.norm(e[:nh]) / np.linalg.norm(xx), np.linalg.norm(e[nh:]) / np.linalg.norm(xx), abs(fx - fxo) / abs(fxo)))
