#!/usr/bin/env python3
"""GPU probe: milestones of the headline fit (|g|/|x| thresholds vs iteration / seconds) for solver variants.
usage: conv_probe2.py variant:cap[,variant:cap...]   variants: vp (default), joint, vp+precond, joint+precond"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED

Q, N, L = 21, int(os.environ.get("PROBE_N", 50000)), int(os.environ.get("PROBE_L", 300))
EPS = float(os.environ.get("PROBE_EPS", 1e-3))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
cns = {}
for spec in (sys.argv[1] if len(sys.argv) > 1 else "vp:1500").split(","):
    name, cap = spec.split(":")
    t = time.time()
    fit = plm.fit(msa, Q, max_iter=int(cap), epsilon=EPS, want_fij=False, precond="precond" in name,
                  joint="joint" in name)
    reach = {}
    for thr in (1.0, 1e-1, 3e-2, 1e-2, 3e-3, 1e-3, 1e-4):
        hit = [r for r in fit["table"] if r[2] < thr]
        reach["%g" % thr] = (hit[0][0], round(hit[0][1], 2)) if hit else None
    print(name, json.dumps(dict(iters=fit["iters"], evals=fit["n_evals"], status=fit["status_msg"],
                                seconds=round(time.time() - t, 2), opt_seconds=round(fit["seconds"]["optimize"], 2),
                                final_cond=fit["table"][-1][2] if fit["table"] else None, fx=fit["fx"], reach=reach)),
          flush=True)
    cns[name] = fit["cn"]
    np.save("gpurun_out/cn_%s.npy" % name, fit["cn"])
names = list(cns)
for k in range(1, len(names)):
    print("max|dCN| %s vs %s: %.3g" % (names[k], names[0], float(np.abs(cns[names[k]] - cns[names[0]]).max())))
