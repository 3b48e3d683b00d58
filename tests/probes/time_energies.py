#!/usr/bin/env python3
"""Wall time of plm.hamiltonians (statistical energies of N sequences under an L-site model) on the headline
shape, next to the CPU oracle (the reference's loop restated in C) on a sample -- SURVEY.md 8f N2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300)); q = 21
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
rng = np.random.default_rng(1)
hi = rng.normal(size=(L, q)).astype(np.float32)
jij = (0.05 * rng.normal(size=(L * (L - 1) // 2, q, q))).astype(np.float32)
plm.hamiltonians(msa[:512], q, hi, jij)            # warm-up (module load, first launch)
t0 = time.perf_counter(); H = plm.hamiltonians(msa, q, hi, jij); t_gpu = time.perf_counter() - t0
t0 = time.perf_counter(); S = plm.single_mutant_matrix(msa[0], q, hi, jij); t_smm = time.perf_counter() - t0
out = {"N": N, "L": L, "hamiltonians_seconds_incl_transfers": round(t_gpu, 4),
       "sequences_per_second": round(N / t_gpu), "single_mutant_matrix_seconds": round(t_smm, 4)}
if os.environ.get("PLM_CPU", "1") == "1":
    from oracle.oracle import Oracle
    o = Oracle("f64")
    ns = 2000
    x = np.concatenate([hi.ravel(), jij.ravel()]).astype(np.float64)
    t0 = time.perf_counter(); Ho = o.hamiltonians(msa[:ns], q, x); t_cpu = time.perf_counter() - t0
    out["cpu_oracle_sequences_per_second_1_thread"] = round(ns / t_cpu)
    out["max_abs_diff_vs_oracle"] = float(np.abs(H[:ns] - Ho).max())
    out["max_abs_energy"] = float(np.abs(Ho).max())
print(out)
