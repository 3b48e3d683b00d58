#!/usr/bin/env python3
"""GPU-side probe: the error of the evaluation at the point a fit stops, with the plain and with the exact forward
GEMM (PLM_FWD_ACCURATE = 0 | 1, read once per context), against the f64 oracle and beside the float32 CPU build.
usage: fwd_accuracy_probe.py [headline|config2|config3|config4|config5 ...] [-g]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
from oracle.oracle import Oracle
CONFIGS = {"config2": (20000, 200, 2), "headline": (50000, 300, 1), "config3": (100000, 300, 3), "config4": (50000, 500, 4),
           "config5": (30000, 600, 5)}
gaps = "-g" in sys.argv
names = [a for a in sys.argv[1:] if a in CONFIGS] or ["headline"]
o64, o32 = Oracle("f64"), Oracle("f32")
ncore = len(os.sched_getaffinity(0))
try:
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
    if quota != "max":
        ncore = min(ncore, max(1, int(int(quota) / int(period))))
except (OSError, ValueError):
    pass
o64.set_num_threads(ncore); o32.set_num_threads(ncore)
Q = 21
for name in names:
    N, L, k = CONFIGS[name]
    msa, _ = synthetic_msa(N, L, seed=BASE_SEED + k)
    lam = plm.default_lambda_j(L, Q - 1 if gaps else Q)
    os.environ.pop("PLM_FWD_ACCURATE", None)
    os.environ.pop("PLM_BWD_PLANES", None)
    with plm.PlmContext(msa, Q, max_iter=3000, epsilon=1e-3, ignore_gaps=gaps) as ctx:
        w, _, neff = ctx.reweight(); ctx.marginals(pairs=False); ctx.set_x(None)
        t = time.time(); r = ctx.optimize(); tf = time.time() - t
        x = ctx.get_x()
    xn = max(1.0, np.linalg.norm(x))
    t = time.time()
    ev = o64.eval_gaps if gaps else o64.eval
    fo, _, go = ev(msa, w.astype(np.float64), Q, 0.01, lam, x.astype(np.float64)); t64 = time.time() - t
    ev32 = o32.eval_gaps if gaps else o32.eval
    _, _, g32 = ev32(msa, w, Q, 0.01, lam, x)
    print("%s%s: fit %d it / %d ev / %.2f s (%s), reported cond %.4g; oracle cond %.4g (%.1f s); f32 CPU err/|x| %.3g" % (
        name, " -g" if gaps else "", r["iters"], r["n_evals"], tf, r["status_msg"][:40], r["table"][-1][2],
        np.linalg.norm(go) / xn, t64, np.linalg.norm(g32.astype(np.float64) - go) / xn), flush=True)
    print("   |x| = %.2f  n_eff = %.0f" % (xn, neff))
    for mode, planes in (("0", "3"), ("1", "3"), ("1", "4")):
        os.environ["PLM_FWD_ACCURATE"] = mode
        os.environ["PLM_BWD_PLANES"] = planes
        with plm.PlmContext(msa, Q, lambda_j=lam, ignore_gaps=gaps) as ctx:
            ctx.set_weights(w); ctx.set_x(x)
            fx, nll = ctx.eval(); g = ctx.get_g()
            t = time.time()
            for _ in range(5):
                ctx.eval()
            tev = (time.time() - t) / 5
        nh = L * (Q - 1 if gaps else Q)
        print("   forward %s: %.2f ms/evaluation  err/|x| total %.3g  fields %.3g  couplings %.3g   fx rel %.2e" % (
            ("exact (%s residual planes)" % planes) if mode == "1" else "plain   ", 1e3 * tev, np.linalg.norm(g - go) / xn, np.linalg.norm((g - go)[:nh]) / xn,
            np.linalg.norm((g - go)[nh:]) / xn, abs(fx - fo) / abs(fo)), flush=True)
    os.environ.pop("PLM_FWD_ACCURATE", None)
    os.environ.pop("PLM_BWD_PLANES", None)
