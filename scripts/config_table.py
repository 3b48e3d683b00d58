#!/usr/bin/env python3
"""BASELINE.json configurations on ONE MI355X: kernel times (HIP events), reweighting, and the whole fit to
|g|/|x| < 1e-3 with the default (variable-projection) solver.  Prints one JSON line per configuration and writes
gpurun_out/config_table.json.  usage: config_table.py [names...]   names: c2 c3 headline c4 c5 headline_g c3_g
(the _g rows: plmc -g / ignore_gaps, the reference's default mode, config/sample_config_monomer.txt:155)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED

CONFIGS = {"c2": (20000, 200, 2), "c3": (100000, 300, 3), "headline": (50000, 300, 1), "c4": (50000, 500, 4),
           "c5": (30000, 600, 5), "headline_g": (50000, 300, 1), "c3_g": (100000, 300, 3)}
names = sys.argv[1:] or list(CONFIGS)
rows = {}
for name in names:
    N, L, k = CONFIGS[name]
    msa, _ = synthetic_msa(N, L, seed=BASE_SEED + k)
    with plm.PlmContext(msa, q=21, max_iter=1000, epsilon=1e-3, ignore_gaps=name.endswith("_g")) as ctx:
        t = time.time(); ctx.reweight(); t_rw = time.time() - t
        ctx.marginals(pairs=False)
        ctx.set_x(None)
        t = time.time(); r = ctx.optimize(); t_fit = time.time() - t
        st = ctx.solver_stats()          # the field solver over the whole fit (HIP events inside the library)
        km = ctx.time_kernels(reps=3)
    rows[name] = dict(N=N, L=L, fit_seconds=round(t_fit, 2), iterations=r["iters"], evaluations=r["n_evals"],
                      status=r["status_msg"], final_cond=r["table"][-1][2], kernel_ms={k: round(v, 3) for k, v in km.items()},
                      field_solver={"ms_per_evaluation": round(st["field_ms_per_evaluation"], 3),
                                    "passes_per_evaluation": round(st["passes_per_evaluation"], 2),
                                    "chains_continued_by_host": st["chains_continued_by_host"]})
    print(name, json.dumps(rows[name]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/config_table.json", "w"), indent=1)
