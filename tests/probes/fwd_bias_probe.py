#!/usr/bin/env python3
"""GPU-side probe: is the error of the forward GEMM's potentials HJ[s,i,a] = sum_j J_ij(a, x_sj) biased?
plm.potentials (the FWD_POTENTIALS epilogue of k_fwd: same K loop as the fit) against a float64 one-hot GEMM in numpy.
A random error averages out over the sequences in the gradient sums, a biased one adds up ~N-fold: the systematic
part of |g_hip - g_f64| at scale (tests/test_gpu_scale.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa
N = int(os.environ.get("PLM_N", 2048)); L = int(os.environ.get("PLM_L", 300)); q = 21
sig = float(os.environ.get("PLM_SIG", 0.1))
msa, _ = synthetic_msa(N, L, seed=5)
rng = np.random.default_rng(1)
npair = L * (L - 1) // 2
jij = (sig * rng.normal(size=(npair, q, q))).astype(np.float32)
hi = np.zeros((L, q), np.float32)
hj = plm.potentials(msa, q, hi, jij).astype(np.float64)             # N x L x q
# float64 reference: dense symmetric coupling matrix [(j,b),(i,a)]
J = np.zeros((L, q, L, q))
iu, ju = np.triu_indices(L, 1)
J[iu, :, ju, :] = jij.astype(np.float64)                             # J[i, a, j, b] = J_ij(a, b)
J[ju, :, iu, :] = np.transpose(jij.astype(np.float64), (0, 2, 1))    # J[j, b, i, a] = J_ij(a, b)
X = np.zeros((N, L, q)); X[np.arange(N)[:, None], np.arange(L)[None, :], msa] = 1.0
ref = (X.reshape(N, L * q) @ J.reshape(L * q, L * q)).reshape(N, L, q)   # sum_{(j,b)} X[s,(j,b)] J[(j,b),(i,a)]
err = hj - ref
mag = np.abs(ref).mean()
print("N=%d L=%d sigma=%.3g: mean|HJ| %.4g  err: mean %.3e  std %.3e  mean(err*sign(HJ)) %.3e  max|err| %.3e" % (
    N, L, sig, mag, err.mean(), err.std(), (err * np.sign(ref)).mean(), np.abs(err).max()))
print("per-(i,a) bias over sequences: rms of column means %.3e (a random error would give %.3e)" % (
    np.sqrt((err.mean(0) ** 2).mean()), err.std() / np.sqrt(N)))
