#!/usr/bin/env python3
"""GPU-side: the default fit of one BASELINE configuration under a list of solver settings, one line per setting:
iterations, evaluations, seconds, field-solver passes per evaluation.  The alignment is built once; every setting gets its
own context (the library reads its environment once per context).
usage: fit_sweep.py CONFIG SETTING [SETTING ...]     CONFIG: c2 headline c3 c4 c5 headline_g
       SETTING: comma-separated NAME=VALUE pairs; m=<int> is the L-BFGS history, everything else an environment variable
       e.g.  fit_sweep.py headline m=6 m=8 m=10 PLM_VP_REL=3e-4 m=10,PLM_VP_REL=3e-4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED

CONFIGS = {"c2": (20000, 200, 2), "c3": (100000, 300, 3), "headline": (50000, 300, 1), "c4": (50000, 500, 4),
           "c5": (30000, 600, 5), "headline_g": (50000, 300, 1)}
name = sys.argv[1]
N, L, k = CONFIGS[name]
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + k)
for setting in sys.argv[2:]:
    m, env = 6, {}
    for kv in setting.split(","):
        key, val = kv.split("=")
        if key == "m":
            m = int(val)
        else:
            env[key] = val
    old = {key: os.environ.get(key) for key in env}
    os.environ.update(env)
    try:
        with plm.PlmContext(msa, q=21, max_iter=1000, epsilon=1e-3, lbfgs_m=m, ignore_gaps=name.endswith("_g")) as ctx:
            ctx.reweight()
            ctx.marginals(pairs=False)
            ctx.set_x(None)
            t = time.time()
            r = ctx.optimize()
            dt = time.time() - t
            st = ctx.solver_stats()
        print("%s %-34s it=%4d ev=%4d %6.3f s  cond=%.2e  passes/ev=%.2f field_ms/ev=%.2f  %s" % (
            name, setting, r["iters"], r["n_evals"], dt, r["table"][-1][2], st["passes_per_evaluation"],
            st["field_ms_per_evaluation"], r["status_msg"][:40]), flush=True)
    finally:
        for key, val in old.items():
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val
