#!/usr/bin/env python3
"""
Runs ON THE GPU BOX: fits a small synthetic alignment with the HIP solver through the
run_plmc boundary and saves (a) the two output files and (b) the raw fit arrays, under
gpurun_out/fixture/.  The files are then validated in the build container by the reference's
own readers (tests/test_reference_pipeline.py) and committed under tests/golden/ as
`hip_fit_L24.*` -- real solver output for the CPU-side plumbing tests.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evcouplings_amd import plm, tools
from evcouplings_amd.synthetic import synthetic_msa, msa_to_a2m

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fixture")
os.makedirs(out, exist_ok=True)
N, L, Q = 500, 24, 21
msa, planted = synthetic_msa(N, L, seed=2024)
ali = msa_to_a2m(msa, os.path.join(out, "hip_fit_L24.a2m"), region_start=10)
res, fit, log = tools.infer_to_files(ali, os.path.join(out, "hip_fit_L24_ECs.txt"), os.path.join(out, "hip_fit_L24.model"),
                                     focus_seq="SYN/10-33", theta=0.8, scale=1.0, iterations=100, lambda_h=0.01,
                                     lambda_J=plm.default_lambda_j(L, Q), lambda_g=0.0, cpu=1)
open(os.path.join(out, "hip_fit_L24.log"), "w").write(log)
np.savez_compressed(os.path.join(out, "hip_fit_L24.npz"), planted=np.array(planted),
                    table=np.array(fit["table"]), n_eff=fit["n_eff"], iters=fit["iters"], n_evals=fit["n_evals"],
                    status=fit["status"], status_msg=np.array(fit["status_msg"]), fx=fit["fx"],
                    lambda_j=fit["lambda_j"], **{k: fit[k] for k in ("weights", "fi", "fij", "hi", "jij", "fn", "cn")})
print("fixture written:", sorted(os.listdir(out)), "iters", fit["iters"], "status", fit["status_msg"])
